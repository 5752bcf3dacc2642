"""CPU: the TORCH_LIBRARY form of the boundary (csrc/torch_ops.cpp) loads without a GPU, registers every op, and
refuses CPU tensors / wrong dtypes / wrong shapes at the op boundary (TORCH_CHECK -> RuntimeError)."""
import pytest
import torch

from nfopp import torch_ops


def test_ops_are_registered():
    ops = torch_ops.load()
    for name in torch_ops.OPS:
        assert hasattr(ops, name), name
    schema = str(torch.ops.nfopp.traj_step.default._schema)
    assert "Tensor(a!) traj" in schema and "float[] hyper" in schema


def test_cpu_tensors_are_refused_at_the_op_boundary():
    ops = torch_ops.load()
    p, x = torch.zeros(33161), torch.zeros(5, 3)
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.onf_fwd_bwd_input(p, x, 0.0, 1.0, True, True, 10)
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.reparametrize(torch.zeros(1, 4, 3), torch.zeros(1, 3), torch.zeros(1, 3), None, None, torch.zeros(4), None)
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.adam_step(torch.zeros(4), torch.zeros(4), torch.zeros(4), torch.zeros(4), 0.9, 0.1, 0.1, 1e-8, 0.01, 1.0)


def test_hyper_list_follows_the_struct_layout():
    import nfopp
    hp = nfopp.TrajectoryHyper(collision_weight=3, bounds=(1, 2, 3, 4), lr=0.02, betas=(0.9, 0.95)).to_c(5)
    h = torch_ops.hyper_list(hp)
    assert len(h) == 18 and h[0] == 3.0 and h[8:12] == [1.0, 2.0, 3.0, 4.0]
    assert abs(h[12] - 0.95) < 1e-7 and abs(h[16] - 0.02 / (1 - 0.9 ** 5)) < 1e-8
