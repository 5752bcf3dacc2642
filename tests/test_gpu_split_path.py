"""GPU: the bf16x3 split-precision matrix path of the fused ONF kernel (csrc/onf_split.hip) against the fp32-MFMA
path, the golden vectors of the reference and a float64 evaluation of the same network.

The split is exact (x = hi + mid + lo) and keeps the six partial products above 2^-24, so the two paths may differ by
accumulation-order rounding only: the gate between them is 3e-6 of the output scale, and BOTH must meet the
reference-parity gates of test_gpu_parity.py."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu

gc = pytest.importorskip("gpu_common")
import nfopp  # noqa: E402
from nfopp import _lib  # noqa: E402
from oracle import nfopp_oracle as orc  # noqa: E402

F32 = np.float32


@pytest.fixture
def split_path():
    lib = _lib.load()
    before = lib.nfopp_get_matrix_path()
    _lib.check(lib.nfopp_set_matrix_path(1))
    assert lib.nfopp_get_matrix_path() == 1
    yield lib
    _lib.check(lib.nfopp_set_matrix_path(before))   # the switch is process-wide: leave it as it was found


def _eval(onf, x, path):
    lib = _lib.load()
    before = lib.nfopp_get_matrix_path()
    _lib.check(lib.nfopp_set_matrix_path(path))
    try:
        out = onf.forward_with_grad(torch.tensor(np.ascontiguousarray(x, F32), device="cuda"))
        logits = onf(torch.tensor(np.ascontiguousarray(x, F32), device="cuda"))
        torch.cuda.synchronize()
        return out.cpu().numpy(), logits.cpu().numpy().reshape(-1)
    finally:
        _lib.check(lib.nfopp_set_matrix_path(before))


def _forward64(params, cfg, x):
    """float64 evaluation of the same network (weights are the fp32 parameters): the yardstick for both paths"""
    p = orc.unpack_params(np.asarray(params, F32), cfg)
    p = {k: (np.asarray(v, np.float64) if v is not None else None) for k, v in p.items()}
    x = np.asarray(x, np.float64)
    u = (x[:, :2] - cfg.mean) / cfg.sigma
    e = u @ p["we"].T + (p["be"] if p["be"] is not None else 0.0)
    feats = [np.sin(e[:, :100])]
    if cfg.use_cos:
        feats.append(np.cos(e[:, 100:]))
    if cfg.angle_encoding:
        z = (x[:, 2:3] + p["ang_b"][None]) * p["ang_f"][None]
        feats += [np.sin(z[:, :cfg.angle_dim]), np.cos(z[:, cfg.angle_dim:])]
    fin = np.concatenate(feats, 1)
    h1 = np.maximum(fin @ p["w1"].T + p["b1"], 0)
    h2 = np.maximum(h1 @ p["w2"].T + p["b2"], 0)
    return np.concatenate([h2, fin], 1) @ p["w3"].reshape(-1) + p["b3"].reshape(-1)[0]


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_split_vs_fp32_path_and_reference(tag):
    z = load_golden("g1_onf.npz")
    onf, cfg = gc.make_onf(z[tag + "_cfg"], z[tag + "_params"])
    x = z[tag + "_x"]
    d = x.shape[1]
    for n in (len(x), 1, 17, 255, 4099, 70000):      # one tile per wave below ~65k points, two above
        rng = np.random.default_rng(n)
        xs = x if n == len(x) else x[rng.integers(0, len(x), n)]
        o0, l0 = _eval(onf, xs, 0)
        o1, l1 = _eval(onf, xs, 1)
        assert gc.scaled_err(o1[:, 0], o0[:, 0]) < 3e-6, (tag, n)
        assert gc.scaled_err(o1[:, 1:1 + d], o0[:, 1:1 + d]) < 3e-6, (tag, n)
        assert gc.scaled_err(l1, l0) < 3e-6
        assert np.array_equal(l1, o1[:, 0]) or gc.scaled_err(l1, o1[:, 0]) < 1e-6   # forward-only kernel = same logits
        if d == 2:
            assert np.all(o1[:, 3] == 0)
    # reference parity gates (identical to test_gpu_parity.py) on the split path
    o1, _ = _eval(onf, x, 1)
    tol = 3e-5 if tag == "b" else 1e-5
    assert gc.scaled_err(o1[:, 0], z[tag + "_logit"]) < tol
    assert gc.scaled_err(o1[:, 1:1 + d], z[tag + "_grad"]) < 5 * tol


def test_split_path_is_at_least_as_close_to_float64():
    z = load_golden("g1_onf.npz")
    onf, cfg = gc.make_onf(z["a_cfg"], z["a_params"])
    x = z["a_x"]
    want = _forward64(z["a_params"], cfg, x)
    o0, _ = _eval(onf, x, 0)
    o1, _ = _eval(onf, x, 1)
    e0, e1 = gc.scaled_err(o0[:, 0], want), gc.scaled_err(o1[:, 0], want)
    # both carry the fp32 rounding of the encoding arguments (|arg| * 6e-8); the split path must not add to it
    assert e1 < 1e-5 and e1 < 1.5 * e0 + 5e-7, (e0, e1)


@pytest.mark.parametrize("name", ["traj_n100_default.npz", "traj_n256_default.npz"])
def test_split_collision_eval_and_rollout(name, split_path):
    zz = load_golden(name)
    onf, cfg = gc.make_onf(zz["cfg"], zz["params"])
    hp = orc.Hyper.from_npz(zz)
    s = gc.state_of(zz, "s0_")
    eng = gc.engine_from_state(onf, s, hp)
    eng.collision_eval(zz["g2_t"][None])
    torch.cuda.synchronize()
    out = eng.onf_out.cpu().numpy()[0]
    assert gc.scaled_err(out[:, 0], zz["g2_logit"]) < 1e-5
    # a batch large enough for two tiles per wave, device Philox draws: both paths from the same state
    B = 300
    sb = gc.state_of(zz, "s0_", reps=B)
    res = []
    for path in (1, 0):
        _lib.check(split_path.nfopp_set_matrix_path(path))
        e = gc.engine_from_state(onf, sb, hp)
        e.seed = 7
        for _ in range(3):
            e.optimize_trajectory()
        torch.cuda.synchronize()
        res.append(e.traj.cpu().numpy())
    _lib.check(split_path.nfopp_set_matrix_path(1))
    assert np.max(np.abs(res[0] - res[1])) < 2e-5


def test_unfrozen_field_sees_every_kind_of_write():
    """DEFAULT: image reuse is opt-in, an unfrozen ONF rebuilds its pre-split weight image in front of every
    launch, so writes torch's version counter does not see -- `.data` assignments, collectives / foreign kernels writing
    through the raw pointer -- show in the very next evaluation."""
    z = load_golden("g1_onf.npz")
    onf, cfg = gc.make_onf(z["a_cfg"], z["a_params"])
    assert not onf.is_frozen
    x = torch.tensor(z["a_x"], device="cuda")
    base = onf.forward_with_grad(x).clone()
    v0 = onf.flat_parameters._version
    onf.flat_parameters.data[:2000] *= 1.5                         # .data write: the version counter does not move
    getattr(onf.mlp, "0").weight.data.mul_(0.5)
    assert onf.flat_parameters._version == v0
    moved = onf.forward_with_grad(x).clone()
    assert not torch.equal(moved, base)
    # a raw-pointer write from outside torch's bookkeeping (what dist.broadcast / all_reduce into the buffer amount to)
    src = torch.tensor(z["a_params"], device="cuda")
    alias = torch.utils.dlpack.from_dlpack(torch.utils.dlpack.to_dlpack(onf.flat_parameters))   # same memory, own bookkeeping
    assert alias.data_ptr() == onf.flat_parameters.data_ptr()
    alias.copy_(src)
    assert onf.flat_parameters._version == v0
    assert torch.equal(onf.forward_with_grad(x), base)


def test_frozen_field_reuses_its_image_and_follows_announced_writes():
    """ABI 5, opt-in: a FROZEN field's launches reuse the stream's pre-split weight image; in-place torch ops (flat buffer or a
    parameter view) and the library's own Adam step are noticed by themselves, anything else through `mark_modified()`;
    `unfreeze()` returns to rebuild-per-launch."""
    z = load_golden("g1_onf.npz")
    onf, cfg = gc.make_onf(z["a_cfg"], z["a_params"])
    x = torch.tensor(z["a_x"], device="cuda")
    lib = _lib.load()
    rebuilt = onf.forward_with_grad(x).clone()
    with onf.frozen():
        assert onf.is_frozen
        base = onf.forward_with_grad(x).clone()
        assert torch.equal(base, rebuilt)                              # cached image: same bits as a rebuilt one
        assert torch.equal(onf.forward_with_grad(x), base)
        with torch.no_grad():
            getattr(onf.mlp, "0").weight.mul_(1.5)                                   # a parameter VIEW, in place
        moved = onf.forward_with_grad(x).clone()
        assert not torch.equal(moved, base)
        with torch.no_grad():
            onf.flat_parameters.copy_(torch.tensor(z["a_params"], device="cuda"))
        assert torch.equal(onf.forward_with_grad(x), base)
        # a write the counter does not see: announced with mark_modified()
        onf.flat_parameters.data.mul_(1.25)
        onf.mark_modified()
        assert not torch.equal(onf.forward_with_grad(x), base)
        onf.flat_parameters.data.copy_(torch.tensor(z["a_params"], device="cuda"))
        onf.mark_modified()
        assert torch.equal(onf.forward_with_grad(x), base)
        # raw-pointer update by the library itself: the registration is withdrawn by nfopp_adam_step
        g = torch.ones_like(onf.flat_parameters)
        m, v = torch.zeros_like(g), torch.zeros_like(g)
        _lib.check(lib.nfopp_adam_step(_lib.ptr(onf.flat_parameters), _lib.ptr(g), _lib.ptr(m), _lib.ptr(v), onf.n_params, 0.9, 0.1,
                                       0.1, 1e-8, 0.05, 1.0, _lib.stream_ptr()))
        after = onf.forward_with_grad(x).clone()
        torch.cuda.synchronize()
        assert not torch.equal(after, base)
    assert not onf.is_frozen
    # unfrozen again, and the library holds no version for the buffer any more: an unannounced .data write shows at once
    onf.flat_parameters.data.copy_(torch.tensor(z["a_params"], device="cuda"))
    assert torch.equal(onf.forward_with_grad(x), rebuilt)


def test_field_built_under_inference_mode_evaluates():
    """inference tensors keep no version counter; reading it raised in every config_c().  Such a field works,
    frozen or not (frozen: only mark_modified() announces writes)."""
    z = load_golden("g1_onf.npz")
    with torch.inference_mode():
        onf, cfg = gc.make_onf(z["a_cfg"], z["a_params"])
        x = torch.tensor(z["a_x"], device="cuda")
        a = onf.forward_with_grad(x).clone()
        onf.freeze()
        b = onf.forward_with_grad(x).clone()
        onf.flat_parameters.mul_(1.5)
        onf.mark_modified()
        c = onf.forward_with_grad(x).clone()
    assert onf._torch_version() is None
    assert torch.equal(a, b) and not torch.equal(b, c)
    ref = gc.make_onf(z["a_cfg"], z["a_params"])[0].forward_with_grad(x)
    assert torch.equal(a, ref)


@pytest.mark.parametrize("use_cos,angle,bias", [(True, True, True), (True, False, True), (False, True, False), (False, False, True),
                                                (True, False, False)])
def test_every_feature_dimension_on_every_matrix_path_vs_oracle(use_cos, angle, bias):
    """The four feature dimensions the kernels are instantiated for (F = 220 / 200 / 120 / 100) with and without the encoding
    bias -- the golden networks cover only 220 and 100 -- on the three matrix paths, both workgroup shapes of the 32x32 kernel
    (ragged counts below and above one 256-sample chunk per CU), against the numpy oracle (fp32 gates of test_gpu_parity.py)."""
    torch.random.manual_seed(11)
    onf = nfopp.ONF(0.4, 2.5, use_cos=use_cos, use_normal_init=True, bias=bias, angle_encoding=angle).to("cuda")
    cfg = orc.OnfConfig(0.4, 2.5, use_cos, bias, angle)
    flat = onf.flat_parameters.cpu().numpy()
    rng = np.random.default_rng(3)
    d = 3 if angle else 2
    for n in (1, 33, 300, 4099, 70001):
        x = rng.uniform(-4, 6, (n, d)).astype(F32)
        if angle:
            x[:, 2] = rng.uniform(-3.3, 3.3, n)
        lo, go = orc.onf_forward_grad(flat, cfg, x)
        res = {}
        for path in (1, 2, 0):
            o, l = _eval(onf, x, path)
            assert gc.scaled_err(o[:, 0], lo) < 1e-5, (path, n)
            assert _kink_free_err(o[:, 1:1 + d], go) < 5e-5, (path, n)
            assert np.array_equal(l, o[:, 0]) or gc.scaled_err(l, o[:, 0]) < 1e-6
            if d == 2:
                assert np.all(o[:, 3] == 0)
            res[path] = o
        for path in (1, 2):
            assert gc.scaled_err(res[path][:, 0], res[0][:, 0]) < 3e-6
            assert _kink_free_err(res[path][:, 1:1 + d], res[0][:, 1:1 + d]) < 3e-6


def _kink_free_err(got, want, allowed=3e-4):
    """Scaled gradient error over all rows but the few that sit ON a ReLU kink: with ~200 hidden units and distinct random
    points, about 1 point in 25 000 has a pre-activation within fp32 rounding of zero, where the derivative of relu jumps --
    two correct fp32 evaluations with different summation orders then differ by one hidden unit's whole contribution (~1 % of
    the gradient) while their logits agree to rounding (measured: the SAME rows in every build and process, tools/x32/
    debug_dims.py).  Those rows (at most `allowed` of them) are set aside; every other row must meet the gate."""
    want = np.asarray(want, np.float64)
    err = np.abs(np.asarray(got, np.float64) - want).max(1) / (np.abs(want).max() + 1e-12)
    worst = np.sort(err)[::-1]
    k = int(allowed * len(err))          # 0 below 3 334 rows: every row is gated
    return float(worst[k])


def _fit_grad(onf, x, y, matrix_path):
    """gradient of the mean BCE loss through the at-scale path (pass 1 + weight-gradient GEMMs) on one matrix path"""
    lib = _lib.load()
    before = lib.nfopp_get_matrix_path()
    _lib.check(lib.nfopp_set_matrix_path(matrix_path))
    try:
        P = x.shape[0]
        c = onf.config_c()
        need = lib.nfopp_onf_train_workspace_bytes(c, P)
        ws = torch.empty((need + 3) // 4, dtype=torch.float32, device="cuda")
        grad = torch.zeros(onf.n_params + 2, device="cuda")
        xd, yd = torch.tensor(x, device="cuda"), torch.tensor(y, device="cuda")
        _lib.check(lib.nfopp_onf_train_grad_ex(c, _lib.ptr(onf.flat_parameters), _lib.ptr(xd), _lib.ptr(yd), P, 1.0 / P,
                                               _lib.ptr(grad), _lib.ptr(ws), ws.numel() * 4, 2, _lib.stream_ptr()))
        torch.cuda.synchronize()
        return grad.cpu().numpy()
    finally:
        _lib.check(lib.nfopp_set_matrix_path(before))


@pytest.mark.parametrize("use_cos,angle,bias", [(True, True, True), (True, False, True), (False, True, False), (False, False, True)])
def test_fit_gradient_every_feature_dimension_on_every_matrix_path_vs_oracle(use_cos, angle, bias):
    """The fit's gradient at scale for F = 220 / 200 / 120 / 100 (the golden networks cover 220 and 100) on the three matrix
    paths -- path 1 runs pass 1 on the 32x32 kernel with the factors stored by index (an odd and an even number of input
    blocks, both workgroup shapes), paths 2 and 0 the 16x16 kernels with their slot orders -- against the numpy oracle
    (nerf_opt_planner.py:83-89 restated), and bit for bit against a repeat."""
    torch.random.manual_seed(5)
    onf = nfopp.ONF(0.4, 2.5, use_cos=use_cos, use_normal_init=True, bias=bias, angle_encoding=angle).to("cuda")
    cfg = orc.OnfConfig(0.4, 2.5, use_cos, bias, angle)
    flat = onf.flat_parameters.cpu().numpy()
    rng = np.random.default_rng(17)
    d = 3 if angle else 2
    for P in (33, 4099, 70001):     # two partly filled tiles / below / above one 256-sample chunk per CU
        x = rng.uniform(-4, 6, (P, d)).astype(F32)
        if angle:
            x[:, 2] = rng.uniform(-3.3, 3.3, P)
        y = (rng.uniform(size=P) < 0.35).astype(F32)
        loss, _, gref = orc.onf_train_grads(flat, cfg, x, y)
        scale = max(1.0, float(np.abs(gref).max()))
        res = {}
        for path in (1, 2, 0):
            g = _fit_grad(onf, x, y, path)
            assert g[-1] == P
            assert abs(float(g[-2]) - float(loss)) < 5e-6 * max(1.0, abs(float(loss))), (path, P)
            assert np.abs(g[:-2] - gref).max() < 2e-5 * scale, (path, P)
            assert np.array_equal(g, _fit_grad(onf, x, y, path)), (path, P)
            res[path] = g
        for path in (1, 2):
            assert np.abs(res[path][:-2] - res[0][:-2]).max() < 2e-5 * scale


@pytest.mark.parametrize("use_cos,angle", [(True, True), (False, False)])
def test_fit_pass1_factors_on_the_32x32_kernel_vs_oracle(use_cos, angle):
    """White box: what pass 1 of the fit leaves in the workspace on matrix path 1 -- the contract between csrc/onf_x32_impl.h's
    training mode and csrc/onf_wgrad.hip (WgradArgs::x32_order): rows  h1 [P,112] | rho*dh1 [P,112] | rho*de [P,16*NKB] |
    record [P,12]  indexed by hidden unit / input feature, ones unit 101, rho row 100, sign bit of a2[s] in word (s>>2)&3 at
    bit 4*(s>>4) + (s&3) -- against the oracle's intermediates (nerf_opt_planner.py:83-89 unrolled), for an even (F = 220) and
    an odd (F = 100) number of input blocks and a ragged sample count."""
    torch.random.manual_seed(9)
    bias = True
    onf = nfopp.ONF(0.4, 2.5, use_cos=use_cos, use_normal_init=True, bias=bias, angle_encoding=angle).to("cuda")
    cfg = orc.OnfConfig(0.4, 2.5, use_cos, bias, angle)
    flat = onf.flat_parameters.cpu().numpy()
    rng = np.random.default_rng(23)
    d, P = (3 if angle else 2), 4099
    x = rng.uniform(-4, 6, (P, d)).astype(F32)
    if angle:
        x[:, 2] = rng.uniform(-3.3, 3.3, P)
    y = (rng.uniform(size=P) < 0.35).astype(F32)
    lib = _lib.load()
    before = lib.nfopp_get_matrix_path()
    _lib.check(lib.nfopp_set_matrix_path(1))
    try:
        c = onf.config_c()
        need = lib.nfopp_onf_train_workspace_bytes(c, P)
        ws = torch.zeros((need + 3) // 4, dtype=torch.float32, device="cuda")
        grad = torch.zeros(onf.n_params + 2, device="cuda")
        xd, yd = torch.tensor(x, device="cuda"), torch.tensor(y, device="cuda")
        _lib.check(lib.nfopp_onf_train_grad_ex(c, _lib.ptr(onf.flat_parameters), _lib.ptr(xd), _lib.ptr(yd), P, 1.0 / P,
                                               _lib.ptr(grad), _lib.ptr(ws), ws.numel() * 4, 2, _lib.stream_ptr()))
        torch.cuda.synchronize()
    finally:
        _lib.check(lib.nfopp_set_matrix_path(before))
    w = ws.cpu().numpy()
    fin = (200 if use_cos else 100) + (20 if angle else 0)
    win = 16 * ((fin + 16) // 16)
    h1 = w[:P * 112].reshape(P, 112)
    dh1 = w[P * 112:2 * P * 112].reshape(P, 112)
    de = w[2 * P * 112:2 * P * 112 + P * win].reshape(P, win)
    rec = w[2 * P * 112 + P * win:2 * P * 112 + P * win + P * 12].reshape(P, 12)
    p = orc.unpack_params(flat, cfg)
    logit, cch = orc._onf_forward_cache(p, cfg, x)
    rho = ((orc.sigmoid(logit) - y) / F32(P)).astype(F32)
    w3 = p["w3"][0]
    dh2 = (rho[:, None] * w3[None, :100] * (cch["a2"] > 0)).astype(F32)
    dh1_o = ((dh2 @ p["w2"]) * (cch["a1"] > 0)).astype(F32)
    din = (dh1_o @ p["w1"] + rho[:, None] * w3[None, 100:]).astype(F32)
    e = cch["e"]
    de_o = np.concatenate([din[:, :100] * np.cos(e[:, :100]), -din[:, 100:200] * np.sin(e[:, 100:])], 1) if use_cos \
        else din[:, :100] * np.cos(e)
    if angle:
        z, k = cch["z"], cfg.angle_dim
        de_o = np.concatenate([de_o, din[:, cfg.n_enc:cfg.n_enc + k] * np.cos(z[:, :k]), -din[:, cfg.n_enc + k:] * np.sin(z[:, k:])], 1)

    def close(got, want, tol):
        return np.abs(got - want).max() <= tol * max(1e-30, float(np.abs(want).max()))

    # rows whose pre-activations sit on a ReLU kink may differ by a whole unit between two correct evaluations: leave out
    # the samples with any |a1| or |a2| below 1e-5 (a handful of 4099)
    ok = (np.abs(cch["a1"]).min(1) > 1e-5) & (np.abs(cch["a2"]).min(1) > 1e-5)
    assert ok.sum() > 0.98 * P
    assert close(h1[:, :100], cch["h1"], 3e-6) and np.all(h1[:, 101] == 1.0)
    assert close(rec[:, 4], rho, 3e-6) and close(dh1[:, 100], rho, 3e-6)
    assert close(dh1[ok, :100], dh1_o[ok], 2e-5)
    assert close(de[ok, :fin], de_o[ok], 2e-5)
    assert np.all(de[:, fin + 1:] == 0.0)                       # pad positions
    u = cch["u"]
    assert np.array_equal(rec[:, 0], u[:, 0]) and np.array_equal(rec[:, 1], u[:, 1]) and np.all(rec[:, 2] == 1.0)
    assert np.array_equal(rec[:, 3], x[:, 2] if angle else np.zeros(P, F32))
    words = rec[:, 8:12].copy().view(np.uint32)
    s = np.arange(100)
    bits = (words[:, (s >> 2) & 3] >> (4 * (s >> 4) + (s & 3))) & 1
    assert np.array_equal(bits[ok].astype(bool), cch["a2"][ok] > 0)
    assert not (words >> 28).any()                              # nothing above the 28 positions of a word
