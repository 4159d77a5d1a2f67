"""Encoder tail: one HIP kernel for s2:34 (L2-normalise) + s4:186 (un-permute) + s4:189 (layout),
and its backward, against the same three statements in torch fp32/fp64."""
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rows,D", [(20, 16), (640, 256), (96, 64), (33, 7), (12, 1030), (5, 1024), (64, 2048)])
@pytest.mark.parametrize("identity", [False, True])
def test_normalize_unperm_forward_and_backward(rows, D, identity):
    from speaker_embedding_ge2e_loss_amd import functional as GF
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(rows * 1000 + D)
    y = (torch.randn(rows, D, generator=g) * 3.0)
    up = torch.randn(rows, D, generator=g)
    perm = random.Random(rows).sample(range(rows), rows)
    unperm = [0] * rows
    for i, j in enumerate(perm):
        unperm[j] = i
    idx = None if identity else unperm

    yr = y.double().requires_grad_(True)
    er = yr / torch.norm(yr, dim=1).unsqueeze(1)          # s2:34
    if idx is not None:
        er = er[idx]                                      # s4:186
    (er * up.double()).sum().backward()

    yd = y.to(dev).requires_grad_(True)
    e = GF.normalize_unperm(yd, idx)
    (e * up.to(dev)).sum().backward()
    assert e.shape == (rows, D)
    assert np.allclose(e.detach().cpu().numpy(), er.detach().numpy(), rtol=2e-6, atol=2e-7)
    assert np.allclose(yd.grad.cpu().numpy(), yr.grad.numpy(), rtol=2e-5, atol=2e-6)


def test_shape_argument_and_bad_permutations():
    from speaker_embedding_ge2e_loss_amd import functional as GF
    dev = torch.device("cuda:0")
    y = torch.randn(12, 8, device=dev)
    e = GF.normalize_unperm(y, list(range(11, -1, -1)), shape=(3, 4))
    assert e.shape == (3, 4, 8) and e.is_contiguous()
    assert torch.allclose(e.reshape(12, 8), torch.nn.functional.normalize(y, dim=1).flip(0), atol=1e-6)
    with pytest.raises(ValueError):
        GF.normalize_unperm(y, [0] * 12)
    with pytest.raises(ValueError):
        GF.normalize_unperm(y, list(range(11)))
    # an int64 device tensor is accepted too
    e2 = GF.normalize_unperm(y, torch.arange(11, -1, -1, device=dev))
    assert torch.equal(e2, e.reshape(12, 8))
    # s2:34 has no epsilon: a zero row is not finite, exactly like the reference's division
    y0 = y.clone()
    y0[3] = 0
    assert not torch.isfinite(GF.normalize_unperm(y0)[3]).any()
