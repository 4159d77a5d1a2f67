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


@pytest.mark.parametrize("shape", [(2, 16, 256), (4, 5, 256), (4, 5, 16), (3, 8, 64), (6, 2, 128), (2, 10, 192), (1, 4, 32)])
@pytest.mark.parametrize("variant", ["softmax", "contrast"])
@pytest.mark.parametrize("identity", [False, True])
def test_loss_raw_is_normalize_unperm_then_loss(shape, variant, identity):
    """ge2e_loss_raw (ONE launch on the encoder's raw projection: s2:34 + s4:186-189 in the loss kernel's load stage, the
    normalisation's backward + scatter in its store stage) against the same statements in torch fp64 followed by the
    oracle's expand form: loss, dL/dy (in y's own row order), dw, db."""
    from oracle import ge2e_oracle as orc
    from speaker_embedding_ge2e_loss_amd import functional as GF
    N, M, D = shape
    if variant == "contrast" and N == 1:
        pytest.skip("contrast needs another speaker")
    assert GF.raw_supported(N, M, D)
    rows = N * M
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(N * 100 + M * 10 + D)
    y = torch.randn(rows, D, generator=g) * (0.5 + 2.0 * torch.rand(rows, 1, generator=g))   # rows of very different norm
    perm = random.Random(rows + D).sample(range(rows), rows)
    unperm = [0] * rows
    for i, j in enumerate(perm):
        unperm[j] = i
    idx = None if identity else unperm

    yr = y.double().requires_grad_(True)
    er = yr / torch.norm(yr, dim=1).unsqueeze(1)                       # s2:34
    if idx is not None:
        er = er[idx]                                                   # s4:186
    wr = torch.tensor(7.0, dtype=torch.float64, requires_grad=True)
    br = torch.tensor(-2.5, dtype=torch.float64, requires_grad=True)
    lr, _, _ = orc.expand_form_loss(er.reshape(N, M, D), wr, br, variant=variant)   # s4:189, s3:19-30
    (1.7 * lr).backward()

    yd = y.to(dev).requires_grad_(True)
    w = torch.tensor(7.0, device=dev, requires_grad=True)
    b = torch.tensor(-2.5, device=dev, requires_grad=True)
    loss = GF.ge2e_loss_raw(yd, idx, w, b, (N, M), variant=variant)
    (1.7 * loss).backward()
    assert np.allclose(loss.item(), lr.item(), rtol=5e-6, atol=2e-6)   # N = 1: the loss is ~eps, an fp32 difference of O(1) terms
    ref = yr.grad.numpy()
    got = yd.grad.cpu().numpy()
    assert np.linalg.norm(got - ref) <= 2e-5 * np.linalg.norm(ref) + 1e-9
    assert np.allclose(w.grad.item(), wr.grad.item(), rtol=1e-4, atol=1e-5)
    assert np.allclose(b.grad.item(), br.grad.item(), atol=1e-4)


def test_loss_raw_falls_back_outside_the_wave_shapes():
    """A shape the one-launch form does not take (N = 16, M = 6): the same call runs normalize_unperm + the loss."""
    from oracle import ge2e_oracle as orc
    from speaker_embedding_ge2e_loss_amd import functional as GF
    N, M, D = 16, 6, 64
    assert not GF.raw_supported(N, M, D)
    dev = torch.device("cuda:0")
    y = torch.randn(N * M, D, generator=torch.Generator().manual_seed(4)) * 2.0
    e = (y / y.norm(dim=1, keepdim=True)).reshape(N, M, D).numpy()
    ref = orc.closed_form(e, 10.0, -5.0)
    w, b = torch.tensor(10.0, device=dev), torch.tensor(-5.0, device=dev)
    loss = GF.ge2e_loss_raw(y.to(dev), None, w, b, (N, M))
    assert np.allclose(loss.item(), ref["loss"], rtol=2e-5)


def test_tensor_index_that_is_no_permutation_leaves_zeros_not_garbage():
    """A tensor `unperm` cannot be validated without a host sync; the outputs it indexes are zero-initialised and the
    kernels skip entries outside the range, so a bad index gives zero rows / zero gradients where nothing was written --
    never uninitialised memory, never an out-of-range access (hygiene)."""
    from speaker_embedding_ge2e_loss_amd import functional as GF
    dev = torch.device("cuda:0")
    rows, D = 20, 256
    torch.full((1 << 22,), float("nan"), device=dev)        # poison the allocator's free blocks
    y = torch.randn(rows, D, device=dev, requires_grad=True)
    bad = torch.arange(rows, device=dev, dtype=torch.int32)
    bad[3] = rows + 5        # out of range: output row 3 is never written
    bad[7] = -2
    bad[9] = 8               # a repeat: row 9 of y is never read, its gradient is never written
    e = GF.normalize_unperm(y, bad)
    e.sum().backward()
    torch.cuda.synchronize()
    assert bool(torch.isfinite(e).all()) and bool((e[3] == 0).all()) and bool((e[7] == 0).all())
    assert bool(torch.isfinite(y.grad).all()) and bool((y.grad[9] == 0).all())
    with pytest.raises(ValueError):
        GF.check_unperm(bad, rows)
    GF.check_unperm(torch.randperm(rows, device=dev), rows)
    # the one-launch raw entry (wave kernel): same index through the gather / scatter of ge2e_loss_fwd_bwd_raw
    y2 = torch.randn(rows, D, device=dev, requires_grad=True)
    w = torch.tensor(10.0, device=dev, requires_grad=True)
    b = torch.tensor(-5.0, device=dev, requires_grad=True)
    loss = GF.ge2e_loss_raw(y2, bad, w, b, (4, 5))
    loss.backward()
    torch.cuda.synchronize()
    assert bool(torch.isfinite(loss)) and bool(torch.isfinite(y2.grad).all()) and bool((y2.grad[9] == 0).all())
