"""GPU: repeated launches on identical inputs must be bitwise identical and correct.

Guards against intra-kernel races / register hazards (one was found in the 8-wave epilogue: see
ge2e_fused_split.hip).  Small N makes a single tile with many pad rows, the shape that exposed it."""
import numpy as np
import pytest
import torch

from oracle import ge2e_oracle as orc
from test_gpu_parity import run_hip, impls_for

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def GF():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from speaker_embedding_ge2e_loss_amd import functional
    return functional


@pytest.mark.parametrize("shape", [(1, 6, 10, 256), (3, 64, 10, 256), (2, 16, 8, 128), (1, 4, 16, 64),
                                   (2100, 4, 5, 256), (37, 2, 16, 256)])   # the last two: one wave per batch
def test_repeated_launches_are_bitwise_identical(GF, shape):
    E = orc.synth_embeddings(shape, "unit", seed=5)
    ref = orc.closed_form(E, 10.0, -5.0)
    for impl in impls_for(GF, *shape):
        first = None
        for k in range(15):
            o = run_hip(GF, E, 10.0, -5.0, impl=impl)
            if first is None:
                first = o
                assert np.abs(o["dE"] - ref["dE"]).max() < 1e-4 * max(1.0, np.abs(ref["dE"]).max()), impl
            else:
                assert np.array_equal(o["dE"], first["dE"]), f"{impl}: launch {k} differs from launch 0"
                assert np.array_equal(o["loss"], first["loss"]) and np.array_equal(o["dw"], first["dw"])
