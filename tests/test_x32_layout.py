"""CPU: the index maps of the 32x32x16 ONF kernel (csrc/onf_x32_impl.h) without a GPU.

tools/x32/emulate_x32.py emulates the gfx950 MFMA operand / accumulator lane maps and both LDS reads (ds_read_b128 rows,
ds_read_b64_tr_b16 transposed) and runs the kernel's address formulas, image packing, third-level fragment order, the
accumulator-as-next-operand chain and the folded bias / skip / ones rows against a plain MLP; tools/x32/lds_search.py counts
LDS cycles of both read patterns under the bank rules of MI355X_MICROARCH.md.  The kernel's constants are checked against both."""
import importlib.util
import os
import re

import numpy as np

import pytest

from conftest import ROOT


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", "x32", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("fin", [220, 200, 120, 100])
def test_fragment_addresses_chain_and_folded_rows(fin, capsys):
    emu = _load("emulate_x32")
    emu.run(fin)                       # asserts inside: every fragment, every layer, logit and input gradient
    assert "OK" in capsys.readouterr().out


def test_kernel_constants_match_the_emulated_layout_and_are_conflict_free():
    emu = _load("emulate_x32")
    src = open(os.path.join(ROOT, "pytorch-motion-planner_amd", "csrc", "onf_x32_impl.h")).read()
    consts = dict(re.findall(r"constexpr int (RS1|RS2|W1_ROWS|W2_ROWS|W1_ZERO|W2_ZERO|SKIP|ONES) = (\d+)", src.replace(",", ";\nconstexpr int")))
    for name in ("RS1", "RS2", "W1_ROWS", "W2_ROWS", "W1_ZERO", "W2_ZERO", "SKIP", "ONES"):
        assert int(consts[name]) == getattr(emu, name), name
    assert "return (row >> 2) & 3;" in src and "return ((row & 3) << 2) | ((row >> 2) & 3);" in src    # swz1 / swz2
    import io, contextlib
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        search = _load("lds_search")   # runs the search at import (a second of numpy-free loops)
    out = buf.getvalue()
    assert "RS=448 swz=(r>>2)&3: fwd 1x  tr 1x" in out                       # W1 image: both reads conflict-free
    assert "RS=256 swz=(r&3)<<2|(r>>2)&3: fwd 1x  tr 1x" in out              # W2 image
    # and the formulas themselves under the emulator's swizzles
    assert search.check(448, emu.swz1, 128, 224) == (1, 1)
    assert search.check(256, emu.swz2, 128, 112) == (1, 1)


def test_second_layer_output_gradient_from_the_masked_outer_products():
    """The identity csrc/onf_wgrad.hip's gather kernel uses in x32 order: with S[i][j] = sum_p rho_p [a2_p[i] > 0] h1_p[j]
    (the weight-gradient GEMM G2 without W3a, ones column included), sum_p rho_p relu(a2_p)[i] = W2[i] . S[i] + b2[i] S[i][ones],
    dW2 = W3a[:, None] * S and db2 = W3a * S[:, ones] -- h2 never has to be summed where it is formed."""
    rng = np.random.default_rng(1)
    P, H = 500, 100
    w2, b2, w3a = rng.normal(size=(H, H)), rng.normal(size=H), rng.normal(size=H)
    h1 = np.maximum(rng.normal(size=(P, H)), 0)
    rho = rng.normal(size=P) / P
    a2 = h1 @ w2.T + b2
    mask = (a2 > 0).astype(np.float64)
    h1e = np.concatenate([h1, np.ones((P, 1))], 1)
    S = (rho[:, None] * mask).T @ h1e                        # [H, H + 1]
    assert np.allclose((w2 * S[:, :H]).sum(1) + b2 * S[:, H], (rho[:, None] * np.maximum(a2, 0)).sum(0), rtol=1e-10, atol=1e-12)
    dh2 = rho[:, None] * w3a[None] * mask
    assert np.allclose(w3a[:, None] * S[:, :H], dh2.T @ h1, rtol=1e-10, atol=1e-12)
    assert np.allclose(w3a * S[:, H], dh2.sum(0), rtol=1e-10, atol=1e-12)
