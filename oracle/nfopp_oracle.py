"""CPU oracle for the NFOPP inner loop -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A numpy/fp32 restatement, with closed-form gradients and a leading batch axis, of the reference's
planner step (MisterMap/pytorch-motion-planner, PyTorch-CPU + autograd).  It is the checker the HIP path is
compared against; only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it.
The product package (`pytorch-motion-planner_amd/nfopp`) never does.

Pinned: every function below is checked in `tests/test_oracle_golden.py` / `tests/test_path_tools_oracle.py` against
golden vectors produced by running the reference itself in the build container (`tests/golden/make_golden.py`,
fixtures committed); the device-RNG samplers restate the product's Philox stream and are checked in
`tests/test_sampling_oracle.py`.

Reference citations are `file:line` under the reference repo root; `nfop/` abbreviates
`neural_field_optimal_planner/`.

A batch of B trajectories is exactly B independent reference problems sharing one ONF (every loss term of
the reference is a `torch.sum` over the waypoints of its single trajectory).
"""
import numpy as np

F32 = np.float32
PI = F32(np.pi)
TWO_PI = F32(2 * np.pi)
HIDDEN = 100


# ----------------------------------------------------------------------------------------------------------
# small helpers
def wrap_angle(a):
    """nfop/torch_math.py:5-7 -- (a + pi) % (2 pi) - pi with fp32 constants, remainder takes the divisor's sign."""
    a = np.asarray(a, F32)
    return (np.remainder(a + PI, TWO_PI).astype(F32) - PI).astype(F32)


def linspace_f32(start, end, steps):
    """torch.linspace on CPU for fp32: symmetric two-sided formula (start + i*step | end - (steps-1-i)*step),
    fp32 step, each element one fused multiply-add (emulated exactly through float64)."""
    start, end = F32(start), F32(end)
    step = np.float64(F32((end - start) / F32(steps - 1)))
    i = np.arange(steps)
    lo = (np.float64(start) + step * i).astype(F32)
    hi = (np.float64(end) - step * (steps - 1 - i)).astype(F32)
    return np.where(i < steps // 2, lo, hi).astype(F32)


def sigmoid(x):
    x = np.asarray(x, F32)
    return (F32(1) / (F32(1) + np.exp(-x, dtype=F32))).astype(F32)


# ----------------------------------------------------------------------------------------------------------
# ONF (occupancy neural field)
class OnfConfig(object):
    """Shape/normalisation of an ONF instance: nfop/onf_model.py:8-31."""

    def __init__(self, mean=0.0, sigma=1.0, use_cos=True, bias=True, angle_encoding=True, angle_dim=10):
        self.mean, self.sigma = float(mean), float(sigma)
        self.use_cos, self.bias, self.angle_encoding = bool(use_cos), bool(bias), bool(angle_encoding)
        self.angle_dim = int(angle_dim) if angle_encoding else 0
        self.n_enc = 200 if use_cos else 100          # encoding_layer outputs (onf_model.py:15,29)
        self.n_ang = 2 * self.angle_dim                # AngleEncoder.encoding_dimension (angle_encoder.py:20-22)
        self.feature_dim = self.n_enc + self.n_ang
        self.point_dim = 3 if angle_encoding else 2

    @classmethod
    def from_vector(cls, v):
        """[mean, sigma, use_cos, bias, angle_encoding] as stored in the golden fixtures."""
        return cls(v[0], v[1], bool(v[2]), bool(v[3]), bool(v[4]))

    def layout(self):
        """(name, shape) in `state_dict()` order -- the flat parameter buffer order."""
        f, h = self.feature_dim, HIDDEN
        out = []
        if self.angle_encoding:
            out += [("ang_b", (self.n_ang,)), ("ang_f", (self.n_ang,))]
        out += [("w1", (h, f)), ("b1", (h,)), ("w2", (h, h)), ("b2", (h,)), ("w3", (1, h + f)), ("b3", (1,)),
                ("we", (self.n_enc, 2))]
        if self.bias:
            out += [("be", (self.n_enc,))]
        return out

    def n_params(self):
        return int(sum(int(np.prod(s)) for _, s in self.layout()))


def unpack_params(flat, cfg):
    flat = np.asarray(flat, F32)
    assert flat.size == cfg.n_params(), (flat.size, cfg.n_params())
    out, o = {}, 0
    for name, shape in cfg.layout():
        n = int(np.prod(shape))
        out[name] = flat[o:o + n].reshape(shape)
        o += n
    if not cfg.bias:
        out["be"] = np.zeros(cfg.n_enc, F32)
    return out


def pack_params(p, cfg):
    return np.concatenate([np.asarray(p[name], F32).reshape(-1) for name, _ in cfg.layout()]).astype(F32)


def _onf_forward_cache(p, cfg, x):
    """nfop/onf_model.py:33-50, nfop/angle_encoder.py:15-18."""
    x = np.asarray(x, F32)
    u = ((x[:, :2] - F32(cfg.mean)) / F32(cfg.sigma)).astype(F32)
    e = (u @ p["we"].T + p["be"]).astype(F32)
    if cfg.use_cos:
        s = np.concatenate([np.sin(e[:, :100]), np.cos(e[:, 100:])], 1).astype(F32)
    else:
        s = np.sin(e).astype(F32)
    c = {"u": u, "e": e}
    if cfg.angle_encoding:
        d = cfg.angle_dim
        z = ((x[:, 2:3] + p["ang_b"][None]) * p["ang_f"][None]).astype(F32)
        code = np.concatenate([np.sin(z[:, :d]), np.cos(z[:, d:])], 1).astype(F32)
        inp = np.concatenate([s, code], 1)
        c["z"] = z
    else:
        inp = s
    a1 = (inp @ p["w1"].T + p["b1"]).astype(F32)
    h1 = np.maximum(a1, 0)
    a2 = (h1 @ p["w2"].T + p["b2"]).astype(F32)
    h2 = np.maximum(a2, 0)
    logit = (np.concatenate([h2, inp], 1) @ p["w3"].T + p["b3"]).astype(F32)[:, 0]
    c.update(inp=inp, a1=a1, h1=h1, a2=a2, h2=h2)
    return logit, c


def onf_forward(flat, cfg, x):
    return _onf_forward_cache(unpack_params(flat, cfg), cfg, x)[0]


def _onf_input_backward(p, cfg, x, c, rho=None):
    """Closed-form d logit / d(x, y[, theta]).  `rho` = optional upstream per point."""
    h = HIDDEN
    w3 = p["w3"][0]
    dh2 = (w3[None, :h] * (c["a2"] > 0)).astype(F32)
    dh1 = ((dh2 @ p["w2"]) * (c["a1"] > 0)).astype(F32)
    din = (dh1 @ p["w1"] + w3[None, h:]).astype(F32)
    e = c["e"]
    if cfg.use_cos:
        de = np.concatenate([din[:, :100] * np.cos(e[:, :100]), -din[:, 100:200] * np.sin(e[:, 100:])], 1)
    else:
        de = din[:, :100] * np.cos(e)
    de = de.astype(F32)
    gxy = ((de @ p["we"]) / F32(cfg.sigma)).astype(F32)
    out = {"dh2": dh2, "dh1": dh1, "din": din, "de": de}
    if cfg.angle_encoding:
        d = cfg.angle_dim
        z, f = c["z"], p["ang_f"]
        dz = np.concatenate([din[:, cfg.n_enc:cfg.n_enc + d] * np.cos(z[:, :d]),
                             -din[:, cfg.n_enc + d:] * np.sin(z[:, d:])], 1).astype(F32)
        gth = np.sum(dz * f[None], 1, dtype=F32)
        out["dz"] = dz
        g = np.concatenate([gxy, gth[:, None]], 1)
    else:
        g = gxy
    return g.astype(F32), out


def onf_forward_grad(flat, cfg, x):
    """-> (logit[P], dlogit/dx [P, point_dim])."""
    p = unpack_params(flat, cfg)
    logit, c = _onf_forward_cache(p, cfg, x)
    g, _ = _onf_input_backward(p, cfg, x, c)
    return logit, g


def onf_train_grads(flat, cfg, x, labels):
    """BCE-with-logits (mean) loss and flat parameter gradient: nfop/nerf_opt_planner.py:83-89 (all parameters
    incl. the angle frequencies receive gradients because of `requires_grad_(True)`, SURVEY quirk v)."""
    p = unpack_params(flat, cfg)
    x = np.asarray(x, F32)
    y = np.asarray(labels, F32)
    P = x.shape[0]
    logit, c = _onf_forward_cache(p, cfg, x)
    loss = np.mean(np.maximum(logit, 0) - logit * y + np.log1p(np.exp(-np.abs(logit))), dtype=np.float64)
    rho = ((sigmoid(logit) - y) / F32(P)).astype(F32)
    h = HIDDEN
    w3 = p["w3"][0]
    g = {}
    cat = np.concatenate([c["h2"], c["inp"]], 1)
    g["w3"] = (rho[None] @ cat).astype(F32)
    g["b3"] = np.asarray([rho.sum(dtype=F32)], F32)
    dh2 = (rho[:, None] * w3[None, :h] * (c["a2"] > 0)).astype(F32)
    g["w2"] = (dh2.T @ c["h1"]).astype(F32)
    g["b2"] = dh2.sum(0, dtype=F32)
    dh1 = ((dh2 @ p["w2"]) * (c["a1"] > 0)).astype(F32)
    g["w1"] = (dh1.T @ c["inp"]).astype(F32)
    g["b1"] = dh1.sum(0, dtype=F32)
    din = (dh1 @ p["w1"] + rho[:, None] * w3[None, h:]).astype(F32)
    e = c["e"]
    if cfg.use_cos:
        de = np.concatenate([din[:, :100] * np.cos(e[:, :100]), -din[:, 100:200] * np.sin(e[:, 100:])], 1)
    else:
        de = din[:, :100] * np.cos(e)
    de = de.astype(F32)
    g["we"] = (de.T @ c["u"]).astype(F32)
    g["be"] = de.sum(0, dtype=F32)
    if cfg.angle_encoding:
        d = cfg.angle_dim
        z = c["z"]
        dz = np.concatenate([din[:, cfg.n_enc:cfg.n_enc + d] * np.cos(z[:, :d]),
                             -din[:, cfg.n_enc + d:] * np.sin(z[:, d:])], 1).astype(F32)
        g["ang_b"] = (dz * p["ang_f"][None]).sum(0, dtype=F32)
        g["ang_f"] = (dz * (x[:, 2:3] + p["ang_b"][None])).sum(0, dtype=F32)
    return F32(loss), logit, pack_params(g, cfg)


def adam_update(param, grad, m, v, step, lr, beta1, beta2, eps):
    """torch.optim.Adam single-tensor path (torch 2.x): lerp / addcmul / addcdiv.  `step` is the 1-based count
    AFTER increment.  Scalars are formed in Python doubles and applied in fp32, as torch does."""
    param, grad, m, v = (np.asarray(a, F32) for a in (param, grad, m, v))
    m = (m + (grad - m) * F32(1 - beta1)).astype(F32)
    v = (v * F32(beta2) + (grad * grad) * F32(1 - beta2)).astype(F32)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    step_size = lr / bc1
    denom = (np.sqrt(v) / F32(bc2 ** 0.5) + F32(eps)).astype(F32)
    param = (param - F32(step_size) * (m / denom)).astype(F32)
    return param, m, v


# ----------------------------------------------------------------------------------------------------------
# trajectory terms (SE(2), ConstrainedNERFOptPlanner)
class Hyper(object):
    """Scalar hyper-parameters of ConstrainedNERFOptPlanner (nfop/constrained_nerf_opt_planner.py:13-40)."""

    def __init__(self, collision_weight=1.0, angle_weight=0.5, constraint_deltas_weight=20.0, multipliers_lr=0.1,
                 collision_multipliers_lr=1e-3, boundary_weight=1.0, collision_beta=1.0,
                 direction_delta_weight=0.0, lr=1e-2, beta1=0.9, beta2=0.9, eps=1e-8,
                 bounds=(-0.1, 3.1, -0.1, 3.1)):
        self.collision_weight = float(collision_weight)
        self.angle_weight = float(angle_weight)
        self.constraint_deltas_weight = float(constraint_deltas_weight)
        self.multipliers_lr = float(multipliers_lr)
        self.collision_multipliers_lr = float(collision_multipliers_lr)
        self.boundary_weight = float(boundary_weight)
        self.collision_beta = float(collision_beta)
        self.direction_delta_weight = float(direction_delta_weight)
        self.lr, self.beta1, self.beta2, self.eps = float(lr), float(beta1), float(beta2), float(eps)
        self.bounds = tuple(float(b) for b in bounds)

    @classmethod
    def from_npz(cls, z, prefix="hp_"):
        kw = {}
        for k in ("collision_weight", "angle_weight", "constraint_deltas_weight", "multipliers_lr",
                  "collision_multipliers_lr", "boundary_weight", "collision_beta", "direction_delta_weight",
                  "lr", "beta1", "beta2", "eps"):
            kw[k] = float(z[prefix + k])
        kw["bounds"] = tuple(float(b) for b in z[prefix + "bounds"])
        return cls(**kw)


def full_trajectory(traj, start, goal):
    """nfop/nerf_opt_planner.py:73-74 with a batch axis: [B,N,D] + [B,D] -> [B,N+2,D]."""
    return np.concatenate([start[:, None], traj, goal[:, None]], 1).astype(F32)


def sample_collision_points(traj, t):
    """nfop/constrained_nerf_opt_planner.py:78-81: between consecutive INTERIOR waypoints, theta wrapped."""
    d = (traj[:, :-1] - traj[:, 1:]).astype(F32)
    d[..., 2] = wrap_angle(d[..., 2])
    return (traj[:, 1:] + t[..., None] * d).astype(F32)


def trajectory_loss_terms(traj, start, goal, lam, cm, t, logit, dlogit, hp):
    """All loss terms of constrained:76-130 + nerf:171-176 and their closed-form gradients.

    traj [B,N,3]; start/goal [B,3]; lam [B,N+1]; cm [B,N]; t [B,N-1]; logit [B,N-1]; dlogit [B,N-1,3]
    (ONF evaluated at `sample_collision_points`).  Returns dict of per-trajectory terms and gradients."""
    traj, start, goal, lam, cm, t = (np.asarray(a, F32) for a in (traj, start, goal, lam, cm, t))
    B, N, _ = traj.shape
    q = full_trajectory(traj, start, goal)
    G = np.zeros_like(q)
    aw = F32(hp.angle_weight)

    # A7 distance (constrained:120-130): raw theta deltas + constant winding correction on the last segment
    delta = (q[:, 1:] - q[:, :-1]).astype(F32)
    wrapped = wrap_angle(delta[..., 2])
    angle_sum = (wrapped.sum(1, dtype=F32) - q[:, -1, 2] + q[:, 0, 2]).astype(F32)
    delta[:, -1, 2] += angle_sum
    delta[..., 2] *= aw
    l_dist = np.sum(delta.astype(F32) ** 2, axis=(1, 2), dtype=F32)
    gd = (F32(2) * delta).astype(F32)
    gd[..., 2] *= aw
    G[:, 1:] += gd
    G[:, :-1] -= gd

    # A5 non-holonomic (constrained:102-109)
    dx = (q[:, 1:, 0] - q[:, :-1, 0]).astype(F32)
    dy = (q[:, 1:, 1] - q[:, :-1, 1]).astype(F32)
    th = q[..., 2]
    m = (th[:, :-1] + wrap_angle(th[:, 1:] - th[:, :-1]) / F32(2)).astype(F32)
    sm, cmm = np.sin(m).astype(F32), np.cos(m).astype(F32)
    c = (dx * sm - dy * cmm).astype(F32)
    e = (dx * cmm + dy * sm).astype(F32)
    l_lin = np.sum(lam * c, 1, dtype=F32)
    l_c2 = np.sum(c * c, 1, dtype=F32)
    g = (lam + F32(2 * hp.constraint_deltas_weight) * c).astype(F32)
    G[:, 1:, 0] += g * sm
    G[:, :-1, 0] -= g * sm
    G[:, 1:, 1] -= g * cmm
    G[:, :-1, 1] += g * cmm
    G[:, :-1, 2] += g * e / F32(2)
    G[:, 1:, 2] += g * e / F32(2)

    # A6 direction / forward-only (constrained:111-118, :93, :98)
    mp = (th[:, :-1] + wrap_angle(th[:, :-1] - th[:, 1:]) / F32(2)).astype(F32)
    smp, cmp_ = np.sin(mp).astype(F32), np.cos(mp).astype(F32)
    d = (-(cmp_ * dx + smp * dy)).astype(F32)
    r = np.where(d > 0, d, F32(0)).astype(F32)
    l_dir = np.sum(r * r, 1, dtype=F32)
    hh = (F32(2 * hp.direction_delta_weight) * r).astype(F32)
    k = (smp * dx - cmp_ * dy).astype(F32)
    G[:, 1:, 0] -= hh * cmp_
    G[:, :-1, 0] += hh * cmp_
    G[:, 1:, 1] -= hh * smp
    G[:, :-1, 1] += hh * smp
    G[:, :-1, 2] += F32(1.5) * hh * k
    G[:, 1:, 2] -= F32(0.5) * hh * k

    # A8 boundary (nerf:171-176), interior only
    lo_x, hi_x, lo_y, hi_y = (F32(b) for b in hp.bounds)
    x, y = traj[..., 0], traj[..., 1]
    bx0, bx1 = np.maximum(lo_x - x, 0), np.maximum(x - hi_x, 0)
    by0, by1 = np.maximum(lo_y - y, 0), np.maximum(y - hi_y, 0)
    l_bnd = np.sum(bx0 ** 2 + bx1 ** 2 + by0 ** 2 + by1 ** 2, 1, dtype=F32)
    wb = F32(2 * hp.boundary_weight)
    G[:, 1:-1, 0] += wb * (bx1 - bx0)
    G[:, 1:-1, 1] += wb * (by1 - by0)

    # A4 collision (constrained:78-89)
    beta = F32(hp.collision_beta)
    bl = (logit * beta).astype(F32)
    with np.errstate(over="ignore"):
        z = np.exp(bl, dtype=F32)
        sp = np.where(bl > 20, logit, np.log1p(z) / beta).astype(F32)
        dsp = np.where(bl > 20, F32(1), z / (z + F32(1))).astype(F32)
    th_l = np.tanh(logit).astype(F32)
    cm_i = (cm[:, 1:] * (F32(1) - t) + cm[:, :-1] * t).astype(F32)
    l_col = np.sum(sp, 1, dtype=F32)
    l_cm = np.sum(cm_i * th_l, 1, dtype=F32)
    gamma = (F32(hp.collision_weight) * dsp + cm_i * (F32(1) - th_l * th_l)).astype(F32)
    gg = (gamma[..., None] * dlogit).astype(F32)
    G[:, 1:-2] += t[..., None] * gg            # a = traj[:-1]  -> full indices 1..N-1
    G[:, 2:-1] += (F32(1) - t)[..., None] * gg  # b = traj[1:]   -> full indices 2..N
    g_cm = np.zeros_like(cm)
    g_cm[:, 1:] += (F32(1) - t) * th_l
    g_cm[:, :-1] += t * th_l

    total = (l_dist + F32(hp.collision_weight) * l_col + l_lin + F32(hp.constraint_deltas_weight) * l_c2
             + F32(hp.boundary_weight) * l_bnd + l_cm + F32(hp.direction_delta_weight) * l_dir).astype(F32)
    return dict(total=total, l_dist=l_dist, l_col=l_col, l_lin=l_lin, l_c2=l_c2, l_bnd=l_bnd, l_cm=l_cm,
                l_dir=l_dir, c=c, d=d, g_traj=G[:, 1:-1].astype(F32), g_lam=c, g_cm=g_cm.astype(F32))


def calculate_inv_hessian(n, w):
    """nfop/nerf_opt_planner.py:45-58: float64 inverse of w*tridiag(-2,4,-2)+I, rounded to fp32."""
    k = np.zeros((n, n), np.float32)
    i = np.arange(n)
    k[i, i] = 4
    k[i[1:], i[:-1]] = -2
    k[i[:-1], i[1:]] = -2
    h = w * k + np.eye(n)
    return np.linalg.inv(h).astype(F32)


def optimize_trajectory(traj, start, goal, lam, cm, adam_m, adam_v, adam_step, t, onf_flat, cfg, hp, hinv):
    """One `_optimize_trajectory` (nerf:143-155 + constrained:63-74) for a batch.  `adam_step` = count BEFORE
    this step.  Returns new (traj, lam, cm, m, v) and the loss terms."""
    B, N, _ = traj.shape
    pts = sample_collision_points(np.asarray(traj, F32), np.asarray(t, F32))
    logit, dl = onf_forward_grad(onf_flat, cfg, pts.reshape(-1, 3))
    terms = trajectory_loss_terms(traj, start, goal, lam, cm, t, logit.reshape(B, N - 1), dl.reshape(B, N - 1, 3), hp)
    g = np.einsum("ij,bjd->bid", hinv, terms["g_traj"]).astype(F32)
    new_traj, m, v = adam_update(traj, g, adam_m, adam_v, adam_step + 1, hp.lr, hp.beta1, hp.beta2, hp.eps)
    new_lam = (np.asarray(lam, F32) + F32(hp.multipliers_lr) * terms["g_lam"]).astype(F32)
    new_cm = (np.asarray(cm, F32) + F32(hp.collision_multipliers_lr) * terms["g_cm"]).astype(F32)
    new_cm = np.where(new_cm > 0, new_cm, F32(0)).astype(F32)
    return new_traj, new_lam, new_cm, m, v, terms


def _torch_row_sum(elems, zero):
    """ATen `row_sum` (native/cpu/SumKernel.cpp): four interleaved accumulator chains over rows of 4 elements,
    folded into the next cascade level every 16 rows (level_power = 4 for fewer than 2^21 elements)."""
    size = len(elems)
    rows = size // 4
    acc = [[zero.copy() for _ in range(4)] for _ in range(4)]
    i = 0
    while i + 16 <= rows:
        for _ in range(16):
            for k in range(4):
                acc[0][k] = (acc[0][k] + elems[i * 4 + k]).astype(F32)
            i += 1
        for j in range(1, 4):
            for k in range(4):
                acc[j][k] = (acc[j][k] + acc[j - 1][k]).astype(F32)
                acc[j - 1][k] = zero.copy()
            if i & (15 << (4 * j)):
                break
    while i < rows:
        for k in range(4):
            acc[0][k] = (acc[0][k] + elems[i * 4 + k]).astype(F32)
        i += 1
    for j in range(1, 4):
        for k in range(4):
            acc[0][k] = (acc[0][k] + acc[j][k]).astype(F32)
    for r in range(rows * 4, size):
        acc[0][0] = (acc[0][0] + elems[r]).astype(F32)
    for k in range(1, 4):
        acc[0][0] = (acc[0][0] + acc[0][k]).astype(F32)
    return acc[0][0]


def torch_sum_f32(a):
    """`torch.sum` of a contiguous fp32 vector on CPU, bit for bit (ATen cascade sum with 8-float vectors -- the
    SumKernel build that torch 2.10 dispatches to, also on AVX-512 hosts): lane columns summed by `_torch_row_sum`,
    then the scalar tail, then the 8 lanes in order.  Checked against torch in tests/test_oracle_golden.py."""
    a = np.asarray(a, F32)
    n = len(a)
    if n >= 8:
        nv = n // 8
        lanes = _torch_row_sum([a[i * 8:(i + 1) * 8] for i in range(nv)], np.zeros(8, F32))
        fin = F32(0)
        for k in range(nv * 8, n):
            fin = F32(fin + a[k])
        for k in range(8):
            fin = F32(fin + lanes[k])
        return fin
    return F32(_torch_row_sum([np.asarray(x, F32) for x in a], np.asarray(0, F32)))


def torch_norm2_f32(dx, dy):
    """`torch.norm(dim=1)` of (dx, dy) rows on CPU: sqrt(fma(dy, dy, rn(dx*dx))) (NormTwoOps, contracted by gcc).
    The fma is emulated in float64: the product and sum of fp32 values there round once more only in ties."""
    dx, dy = np.asarray(dx, F32), np.asarray(dy, F32)
    xx = (dx * dx).astype(F32).astype(np.float64)
    return np.sqrt((xx + dy.astype(np.float64) * dy.astype(np.float64)).astype(F32)).astype(F32)


def _searchsorted_left(cdf, u):
    return np.stack([np.searchsorted(cdf[b], u, side="left") for b in range(cdf.shape[0])])


def reparametrize(traj, start, goal, lam=None, cm=None):
    """Arc-length (xy) reparametrisation: constrained:132-171 (SE(2), with multipliers) / nerf:224-244 (2-D)."""
    traj = np.asarray(traj, F32)
    B, N, D = traj.shape
    q = full_trajectory(traj, np.asarray(start, F32), np.asarray(goal, F32))
    seg = (q[:, 1:, :2] - q[:, :-1, :2]).astype(F32)
    # the cdf feeds searchsorted (index work): torch-CPU roundings restated bit for bit -- torch.norm, torch.sum
    # (cascade order) and torch.cumsum (float64 accumulator, every partial rounded to fp32)
    dist = torch_norm2_f32(seg[..., 0], seg[..., 1])
    total = np.asarray([torch_sum_f32(dist[b]) for b in range(B)], F32)
    nd = (dist / total[:, None]).astype(F32)
    cdf = np.concatenate([np.zeros((B, 1), F32), np.cumsum(nd.astype(np.float64), 1).astype(F32)], 1)
    u = linspace_f32(0, 1, N + 2)[1:-1]
    idx = _searchsorted_left(cdf, u)
    ia = np.where(idx > N + 1, N + 1, idx)
    ib = np.where(idx - 1 < 0, 0, idx - 1)
    ca = np.take_along_axis(cdf, ia, 1)
    cb = np.take_along_axis(cdf, ib, 1)
    den = (ca - cb).astype(F32)
    den = np.where(den < F32(1e-5), F32(1e-5), den)
    tau = ((u[None] - cb) / den).astype(F32)
    qa = np.take_along_axis(q, ia[..., None], 1)
    qb = np.take_along_axis(q, ib[..., None], 1)
    out = np.empty_like(traj)
    if D == 2:
        out[:] = (F32(1) - tau)[..., None] * qb + tau[..., None] * qa
        return out.astype(F32)
    out[..., :2] = (F32(1) - tau)[..., None] * qb[..., :2] + tau[..., None] * qa[..., :2]
    out[..., 2] = qb[..., 2] + tau * wrap_angle(qa[..., 2] - qb[..., 2])
    cm = np.asarray(cm, F32)
    lam = np.asarray(lam, F32)
    cmf = np.concatenate([np.zeros((B, 1), F32), cm, np.zeros((B, 1), F32)], 1)
    new_cm = ((F32(1) - tau) * np.take_along_axis(cmf, ib, 1) + tau * np.take_along_axis(cmf, ia, 1)).astype(F32)
    lf = np.concatenate([lam[:, :1], (lam[:, :-1] + lam[:, 1:]) / F32(2), lam[:, -1:]], 1).astype(F32)
    li = ((F32(1) - tau) * np.take_along_axis(lf, ib, 1) + tau * np.take_along_axis(lf, ia, 1)).astype(F32)
    new_lam = np.concatenate([li[:, :1], (li[:, :-1] + li[:, 1:]) / F32(2), li[:, -1:]], 1).astype(F32)
    return out.astype(F32), new_lam, new_cm


def planner_step(state, t, onf_flat, cfg, hp, hinv, reparam_freq=10):
    """One frozen-ONF `step()` (nerf:60-71): trajectory optimisation, then reparametrisation when
    `step_count % reparam_freq == 0`, then step_count += 1.  `state` is a dict updated in place."""
    tr, lam, cm, m, v, terms = optimize_trajectory(state["traj"], state["start"], state["goal"], state["lam"],
                                                   state["cm"], state["adam_m"], state["adam_v"],
                                                   state["adam_step"], t, onf_flat, cfg, hp, hinv)
    state.update(traj=tr, lam=lam, cm=cm, adam_m=m, adam_v=v, adam_step=state["adam_step"] + 1)
    if state["step_count"] % reparam_freq == 0:
        tr, lam, cm = reparametrize(state["traj"], state["start"], state["goal"], state["lam"], state["cm"])
        state.update(traj=tr, lam=lam, cm=cm)
    state["step_count"] += 1
    return terms


# ----------------------------------------------------------------------------------------------------------
# 2-D planner (NERFOptPlanner)
def trajectory_loss_2d(traj, start, goal, t, logit, dlogit, collision_weight):
    """nfop/nerf_opt_planner.py:157-169 with p = traj[1:](1-t) + traj[:-1] t (:113-117)."""
    traj = np.asarray(traj, F32)
    q = full_trajectory(traj, np.asarray(start, F32), np.asarray(goal, F32))
    G = np.zeros_like(q)
    delta = (q[:, 1:] - q[:, :-1]).astype(F32)
    l_dist = np.sum(delta ** 2, axis=(1, 2), dtype=F32)
    G[:, 1:] += F32(2) * delta
    G[:, :-1] -= F32(2) * delta
    with np.errstate(over="ignore"):
        z = np.exp(logit, dtype=F32)
        sp = np.where(logit > 20, logit, np.log1p(z)).astype(F32)
        dsp = np.where(logit > 20, F32(1), z / (z + F32(1))).astype(F32)
    l_col = np.sum(sp, 1, dtype=F32)
    gg = (F32(collision_weight) * dsp[..., None] * dlogit).astype(F32)
    G[:, 1:-2] += t[..., None] * gg
    G[:, 2:-1] += (F32(1) - t)[..., None] * gg
    total = (l_dist + l_col * F32(collision_weight)).astype(F32)
    return dict(total=total, l_dist=l_dist, l_col=l_col, g_traj=G[:, 1:-1].astype(F32))


def sample_collision_points_2d(traj, t):
    return (traj[:, 1:] * (F32(1) - t)[..., None] + traj[:, :-1] * t[..., None]).astype(F32)


def optimize_trajectory_2d(traj, start, goal, adam_m, adam_v, adam_step, t, onf_flat, cfg, collision_weight,
                           lr, beta1, beta2, eps, hinv):
    B, N, _ = traj.shape
    pts = sample_collision_points_2d(np.asarray(traj, F32), np.asarray(t, F32))
    logit, dl = onf_forward_grad(onf_flat, cfg, pts.reshape(-1, 2))
    terms = trajectory_loss_2d(traj, start, goal, t, logit.reshape(B, N - 1), dl.reshape(B, N - 1, 2), collision_weight)
    g = np.einsum("ij,bjd->bid", hinv, terms["g_traj"]).astype(F32)
    new_traj, m, v = adam_update(traj, g, adam_m, adam_v, adam_step + 1, lr, beta1, beta2, eps)
    return new_traj, m, v, terms


# ----------------------------------------------------------------------------------------------------------
# ground-truth checkers and initialiser (host-side rows of SURVEY section 8: A14, section 8(f) rank 2)
def check_boundaries(xy, boundaries):
    """nfop/collision_checker/collision_checker.py:12-19."""
    if boundaries is None:
        return np.zeros(len(xy), bool)
    return (xy[:, 0] > boundaries[1]) | (xy[:, 0] < boundaries[0]) | (xy[:, 1] > boundaries[3]) | (xy[:, 1] < boundaries[2])


def circle_check(xy, obstacles, radius, boundaries=None):
    """nfop/collision_checker/circle_collision_checker.py:11-14."""
    d = np.linalg.norm(xy[None] - obstacles[:, None], axis=2)
    return np.any(d < radius, axis=0) | check_boundaries(xy, boundaries)


def rectangle_check(poses, obstacles, box, boundaries=None):
    """nfop/collision_checker/rectangle_collision_checker.py:11-26: obstacle points into the robot frame."""
    x, y, th = poses[:, 0], poses[:, 1], poses[:, 2]
    c, s = np.cos(th), np.sin(th)
    ox, oy = obstacles[:, 0][None], obstacles[:, 1][None]
    rx = c[:, None] * (ox - x[:, None]) + s[:, None] * (oy - y[:, None])
    ry = -s[:, None] * (ox - x[:, None]) + c[:, None] * (oy - y[:, None])
    inside = (rx > box[0]) & (rx < box[1]) & (ry > box[2]) & (ry < box[3])
    return np.any(inside, 1) | check_boundaries(poses[:, :2], boundaries)


def initialize_trajectory(start, goal, n):
    """nfop/trajectory_initializer.py:12-29: straight line in xy, theta along the wrapped shortest rotation."""
    start, goal = np.asarray(start, F32), np.asarray(goal, F32)
    out = np.zeros((n, 3), F32)
    out[:, 0] = linspace_f32(start[0], goal[0], n + 2)[1:-1]
    out[:, 1] = linspace_f32(start[1], goal[1], n + 2)[1:-1]
    ga = F32(wrap_angle(goal[2] - start[2]) + start[2])
    out[:, 2] = linspace_f32(start[2], ga, n + 2)[1:-1]
    return out


# ----------------------------------------------------------------------------------------------------------
# batched sample generation for continuous ONF learning (device restatement of nfop/nerf_opt_planner.py:101-141 with a
# counter-based RNG; csrc/sampling.hip) -- same Philox stream, so the checks are exact up to libm rounding
STREAM_T, STREAM_COURSE, STREAM_FINE, STREAM_FIELD, STREAM_KEY = 1, 2, 3, 4, 5


def philox_uniform(seed, ctr_lo, ctr_hi):
    """Philox4x32-10, output word 0 -> uniform [0, 1) with 24 random bits (csrc/common.h philox_uniform)."""
    ctr_lo = np.asarray(ctr_lo, np.uint64)
    ctr_hi = np.broadcast_to(np.asarray(ctr_hi, np.uint64), ctr_lo.shape)
    mask = np.uint64(0xFFFFFFFF)
    c0, c1 = ctr_lo & mask, ctr_lo >> np.uint64(32)
    c2, c3 = ctr_hi & mask, ctr_hi >> np.uint64(32)
    k0, k1 = np.uint64(seed & 0xFFFFFFFF), np.uint64((seed >> 32) & 0xFFFFFFFF)
    m0, m1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
    for _ in range(10):
        p0, p1 = m0 * c0, m1 * c2
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & mask, p1 >> np.uint64(32), p1 & mask
        c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
        k0 = (k0 + np.uint64(0x9E3779B9)) & mask
        k1 = (k1 + np.uint64(0xBB67AE85)) & mask
    return ((c0 >> np.uint64(8)).astype(np.float64) * 2.0 ** -24).astype(F32)


def draw_uniform(seed, traj, idx, offset, stream):
    lo = np.asarray(idx, np.uint64) | (np.uint64(stream) << np.uint64(32))
    hi = (np.uint64(traj) << np.uint64(24)) ^ np.uint64(offset)
    return philox_uniform(seed, lo, hi)


def draw_normal(seed, traj, idx, offset, stream):
    idx = np.asarray(idx, np.uint64)
    u1 = (F32(1) - draw_uniform(seed, traj, 2 * idx, offset, stream)).astype(F32)
    u2 = draw_uniform(seed, traj, 2 * idx + 1, offset, stream)
    return (np.sqrt(F32(-2) * np.log(u1)) * np.cos(TWO_PI * u2)).astype(F32)


def sample_candidates(prev_traj, pool, pool_age, cap, n_field, course_sigma, fine_sigma, angle_sigma, bounds, seed,
                      offset, traj_index_offset=0):
    """-> (cand [B,C,D], cand_age [B,C], samples [B,S,D]) with the pool slot of `samples` left at zero."""
    prev_traj = np.asarray(prev_traj, F32)
    B, N, D = prev_traj.shape
    pool_n = 0 if pool is None else cap
    C, S = cap + N - 1, (N - 1) + cap + n_field
    cand, cage, smp = np.zeros((B, C, D), F32), np.zeros((B, C), F32), np.zeros((B, S, D), F32)
    j = np.arange(N - 1)
    for b in range(B):
        tg = traj_index_offset + b
        if pool_n:
            cand[b, :cap], cage[b, :cap] = pool[b], pool_age[b]
        t = draw_uniform(seed, tg, j, offset, STREAM_T)[:, None]
        pos = (prev_traj[b, 1:] * (F32(1) - t) + prev_traj[b, :-1] * t).astype(F32)
        for d in range(D):
            sc, sf = (course_sigma, fine_sigma) if d < 2 else (angle_sigma, angle_sigma)
            smp[b, :N - 1, d] = pos[:, d] + draw_normal(seed, tg, D * j + d, offset, STREAM_COURSE) * F32(sc)
            cand[b, pool_n:pool_n + N - 1, d] = pos[:, d] + draw_normal(seed, tg, D * j + d, offset, STREAM_FINE) * F32(sf)
        r = np.arange(n_field)
        lo_x, hi_x, lo_y, hi_y = (F32(v) for v in bounds)
        f0 = (N - 1) + cap
        smp[b, f0:, 0] = lo_x + draw_uniform(seed, tg, D * r, offset, STREAM_FIELD) * (hi_x - lo_x)
        smp[b, f0:, 1] = lo_y + draw_uniform(seed, tg, D * r + 1, offset, STREAM_FIELD) * (hi_y - lo_y)
        if D == 3:
            smp[b, f0:, 2] = draw_uniform(seed, tg, D * r + 2, offset, STREAM_FIELD) * TWO_PI
    return cand[:, :pool_n + N - 1], cage[:, :pool_n + N - 1], smp


def resample_pool(cand, cand_age, logits, cap, seed, offset, traj_index_offset=0):
    """Exponential-race weighted sampling without replacement -> (pool [B,cap,D], age [B,cap], chosen indices)."""
    cand, cand_age, logits = np.asarray(cand, F32), np.asarray(cand_age, F32), np.asarray(logits, F32)
    B, C, D = cand.shape
    pool, age, chosen = np.zeros((B, cap, D), F32), np.zeros((B, cap), F32), np.zeros((B, cap), np.int64)
    c = np.arange(C)
    for b in range(B):
        w = (sigmoid(logits[b]) * np.exp(F32(-0.03) * cand_age[b]) + F32(1e-6)).astype(F32)
        u = (F32(1) - draw_uniform(seed, traj_index_offset + b, c, offset, STREAM_KEY)).astype(F32)
        key = (-np.log(u) / w).astype(F32)
        order = np.lexsort((c, key))[:cap]
        chosen[b], pool[b], age[b] = order, cand[b, order], cand_age[b, order] + F32(1)
    return pool, age, chosen


def grid_check(xy, grid, origin_x, origin_y, cell):
    """MapCollisionChecker (notebooks/onf_planner_image_map.ipynb cell 2): float64 index arithmetic on the poses
    (numpy promotes them there), truncating int32 cast, outside [0, cols-1) x [0, rows-1) = collision.  Pinned by
    tests/golden/g16_grid_checker.npz (labels of the notebook's own class)."""
    xy = np.asarray(xy, np.float64)
    rows, cols = grid.shape
    ix = ((xy[:, 0] - origin_x - cell / 2) / cell).astype(np.int32)
    iy = ((xy[:, 1] - origin_y - cell / 2) / cell).astype(np.int32)
    inside = (ix >= 0) & (iy >= 0) & (iy < rows - 1) & (ix < cols - 1)
    out = np.ones(len(xy), bool)
    out[inside] = grid[iy[inside], ix[inside]] > 0
    return out


# ----------------------------------------------------------------------------------------------------------
# path evaluation (csrc/path_eval.hip; loop of scripts/run_bench_mr.py:109-132 with generic densification)
def path_interpolate(traj, start, goal, sub):
    """-> (poses [B, (N+1)*sub + 1, D], xy polyline length [B])."""
    q = full_trajectory(np.asarray(traj, F32), np.asarray(start, F32), np.asarray(goal, F32))
    B, n2, D = q.shape
    p0, p1 = q[:, :-1], q[:, 1:]
    delta = (p1 - p0).astype(F32)
    if D == 3:
        delta[..., 2] = wrap_angle(delta[..., 2])
    u = (np.arange(sub, dtype=F32) / F32(sub))[None, None, :, None]
    poses = (p0[:, :, None, :] + u * delta[:, :, None, :]).astype(F32).reshape(B, (n2 - 1) * sub, D)
    poses = np.concatenate([poses, q[:, -1:]], 1)
    seg = q[:, 1:, :2] - q[:, :-1, :2]
    length = np.sqrt(np.sum(seg * seg, 2, dtype=F32)).astype(F32).sum(1, dtype=F32)
    return poses, length


def path_select_best(labels, length, traj, best_traj, best_length, active=None):
    """Best-path bookkeeping + early stop.  Returns (collides, best_traj, best_length, active)."""
    labels = np.asarray(labels)
    collides = (labels != 0).any(1)
    was_active = np.ones(len(length), bool) if active is None else np.asarray(active, bool)
    improve = was_active & ~collides & (length < best_length)
    best_traj = np.where(improve[:, None, None], traj, best_traj)
    best_length = np.where(improve, length, best_length)
    new_active = None if active is None else was_active & ~(~collides & ~improve)
    return collides, best_traj, best_length, new_active


# ----------------------------------------------------------------------------------------------------------
# heading initialisation along the travel direction (csrc/traj_init.hip)
def initialize_trajectory_directed(start, goal, n):
    """nfop/trajectory_initializer.py:31-45 (`init_angles_with_trajectory=True`): after the straight-line
    initialisation, pull each heading towards atan2 of the central difference of the full path, weighted by a
    0->1->0 ramp (`cat(linspace(0,1,n//2), linspace(1,0,(n+1)//2))`)."""
    start, goal = np.asarray(start, F32), np.asarray(goal, F32)
    tr = initialize_trajectory(start, goal, n)
    full = np.concatenate([start[None], tr, goal[None]]).astype(F32)
    x = (full[2:, 0] - full[:-2, 0]).astype(F32)
    y = (full[2:, 1] - full[:-2, 1]).astype(F32)
    ang = np.arctan2(y, x).astype(F32)
    w = np.concatenate([linspace_f32(0.0, 1.0, n // 2) if n // 2 > 1 else np.zeros(n // 2, F32),
                        linspace_f32(1.0, 0.0, (n + 1) // 2) if (n + 1) // 2 > 1 else np.ones((n + 1) // 2, F32)])
    d = (wrap_angle((ang - tr[:, 2]).astype(F32)) * w).astype(F32)
    tr[:, 2] = (tr[:, 2] + d).astype(F32)
    return tr


# ----------------------------------------------------------------------------------------------------------
# path post-processing for the follower (csrc/path_post.hip)
def _pairwise_sum_f32(a):
    """numpy's float add.reduce order for a contiguous fp32 vector (pairwise summation: 8 running partial sums
    over blocks of at most 128 elements, halves split at a multiple of 8) -- the order `np.sum` uses in
    ros/path_postprocessor.py:29."""
    a = np.asarray(a, F32)
    n = len(a)
    if n < 8:
        s = F32(0.0)            # numpy starts from -0.0; irrelevant for non-negative terms
        for v in a:
            s = F32(s + v)
        return s
    if n <= 128:
        r = [F32(a[j]) for j in range(8)]
        i = 8
        while i < n - (n % 8):
            for j in range(8):
                r[j] = F32(r[j] + a[i + j])
            i += 8
        s = F32(F32(F32(r[0] + r[1]) + F32(r[2] + r[3])) + F32(F32(r[4] + r[5]) + F32(r[6] + r[7])))
        for v in a[i:]:
            s = F32(s + v)
        return s
    n2 = n // 2
    n2 -= n2 % 8
    return F32(_pairwise_sum_f32(a[:n2]) + _pairwise_sum_f32(a[n2:]))


def _quadratic_spline_knots(x):
    """Knot vector scipy's `make_interp_spline(x, y, k=2)` uses for an even degree (scipy 1.15
    interpolate/_bsplines.py, default boundary conditions): data-site midpoints without the first and last one,
    end knots tripled.  interp1d(kind="quadratic") (ros/path_postprocessor.py:50-52) is exactly this spline."""
    mid = (x[1:] + x[:-1]) / 2.0
    return np.concatenate([[x[0]] * 3, mid[1:-1], [x[-1]] * 3])


def _bspline_basis2(t, ell, x):
    """The three quadratic B-spline basis values B_{ell-2..ell}(x) for t[ell] <= x < t[ell+1] (de Boor-Cox)."""
    h = [1.0, 0.0, 0.0]
    for j in (1, 2):
        hh = list(h)
        h[0] = 0.0
        for n in range(1, j + 1):
            xb, xa = t[ell + n], t[ell + n - j]
            if xb == xa:
                h[n] = 0.0
                continue
            w = hh[n - 1] / (xb - xa)
            h[n - 1] += w * (xb - x)
            h[n] = w * (x - xa)
    return h


def _find_interval(t, m, x):
    """Largest ell in [2, m-1] with t[ell] <= x (clamped at both ends: extrapolation uses the end polynomials)."""
    ell = 2
    while ell < m - 1 and x >= t[ell + 1]:
        ell += 1
    return ell


def path_postprocess(path, minimal_distance=0.001, distance_step=0.05):
    """ros/path_postprocessor.py:13-69 for one fp32 path [n, 3] -> float64 [count', 3].

    Mixed precision as numpy evaluates the reference on an fp32 input: filter/segment lengths/cumsum/unfolded
    headings in fp32 (python scalars are weak), the normalised parametrisation, spline and output in float64.
    The quadratic interpolating spline is tridiagonal in its B-spline coefficients (row j touches j-1..j+1), solved
    here by elimination without pivoting (scipy: LAPACK gbsv) -- agreement to rounding."""
    tr = np.asarray(path, F32)
    if len(tr) < 3:
        return tr.copy()
    # _filter_trajectory (:35-44): walk backwards, keep interior poses further than minimal_distance from the last kept
    md = F32(minimal_distance)
    keep = [len(tr) - 1]
    prev = tr[-1]
    for i in range(len(tr) - 2, 0, -1):
        dx, dy = F32(prev[0] - tr[i, 0]), F32(prev[1] - tr[i, 1])
        if np.sqrt(F32(F32(dx * dx) + F32(dy * dy))) > md:
            keep.append(i)
            prev = tr[i]
    keep.append(0)
    tr = tr[keep[::-1]].copy()
    m = len(tr)
    if m < 3:
        raise ValueError("path collapses to fewer than 3 poses: no quadratic spline")
    seg = (tr[1:, :2] - tr[:-1, :2]).astype(F32)
    dist = (np.sqrt((seg[:, 0] * seg[:, 0] + seg[:, 1] * seg[:, 1]).astype(F32)) + F32(1e-6)).astype(F32)
    cum = np.zeros(m, F32)
    acc = F32(0)
    for i in range(m - 1):
        acc = F32(acc + dist[i])
        cum[i + 1] = acc
    param = cum.astype(np.float64) / np.float64(cum[-1])
    total = _pairwise_sum_f32(dist)
    count = int(F32(total / F32(distance_step)))
    # unfold_angles (utils/math.py:38-43) in fp32, first heading added in float64, stored back as fp32
    pi, two_pi = F32(np.pi), F32(2 * np.pi)
    ang = (np.remainder((tr[:, 2] + pi).astype(F32), two_pi).astype(F32) - pi).astype(F32)
    d = (ang[1:] - ang[:-1]).astype(F32)
    d = np.where(d > pi, (d - two_pi).astype(F32), d)
    d = np.where(d < -pi, (d + two_pi).astype(F32), d).astype(F32)
    cs = np.zeros(m, F32)
    acc = F32(0)
    for i in range(m - 1):
        acc = F32(acc + d[i])
        cs[i + 1] = acc
    tr[:, 2] = (np.float64(ang[0]) + cs.astype(np.float64)).astype(F32)
    y = tr.astype(np.float64)
    # interpolating quadratic spline
    t = _quadratic_spline_knots(param)
    lower, diag, upper = np.zeros(m), np.ones(m), np.zeros(m)
    for j in range(1, m - 1):
        ell = max(2, min(j + 1, m - 1))
        b = _bspline_basis2(t, ell, param[j])
        lower[j], diag[j], upper[j] = b[j - 1 - (ell - 2)], b[j - (ell - 2)], b[j + 1 - (ell - 2)]
    c = y.copy()
    dd = diag.copy()
    for j in range(1, m):
        w = lower[j] / dd[j - 1]
        dd[j] -= w * upper[j - 1]
        c[j] -= w * c[j - 1]
    c[m - 1] /= dd[m - 1]
    for j in range(m - 2, -1, -1):
        c[j] = (c[j] - upper[j] * c[j + 1]) / dd[j]
    if count <= 0:
        return np.zeros((0, 3))
    step = 1.0 / (count - 1) if count > 1 else 0.0
    out = np.zeros((count, 3))
    for q in range(count):
        x = 1.0 if (q == count - 1 and count > 1) else q * step
        ell = _find_interval(t, m, x)
        b = _bspline_basis2(t, ell, x)
        out[q] = b[0] * c[ell - 2] + b[1] * c[ell - 1] + b[2] * c[ell]
    # _find_minimal_filter_index (:54-61): drop the leading poses driven in the other direction (first 6 only)
    first = 1
    if count >= 2:
        delta = out[1:, :2] - out[:-1, :2]
        dth = np.remainder(out[1:, 2] - out[:-1, 2] + np.pi, 2 * np.pi) - np.pi
        mean = out[:-1, 2] + dth / 2
        fwd = np.cos(mean) * delta[:, 0] + np.sin(mean) * delta[:, 1] > 0
        other = np.nonzero(fwd != fwd[0])[0]
        if len(other) > 0 and other[0] < 6:
            first = max(int(other[0]), 1)
    return out[first:]
