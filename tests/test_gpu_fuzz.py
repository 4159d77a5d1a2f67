"""Seeded random shapes through impl="auto" (and every implementation that accepts the shape) against the fp64 closed
form: whatever AUTO resolves to -- wave, team, fused_split, tiled, generic -- has to be right at shapes nobody tuned for."""
import os

import numpy as np
import pytest
import torch

from oracle import ge2e_oracle as orc
from test_gpu_parity import check, impls_for, run_hip

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def GF():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from speaker_embedding_ge2e_loss_amd import functional
    return functional


def cases(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        N = int(rng.integers(1, 71))
        M = int(rng.integers(2, 17))
        D = int(rng.choice([4, 20, 64, 96, 128, 192, 200, 256, 257, 264, 320]))
        B = int(rng.integers(1, 600)) if N * M <= 64 and rng.random() < 0.5 else int(rng.integers(1, 12))
        if B * N * M * D > 6e6:      # keep the fp64 oracle quick
            continue
        variant = "contrast" if (rng.random() < 0.3 and N > 1) else "softmax"
        out.append((B, N, M, D, variant, float(rng.uniform(-4, 14)), float(rng.uniform(-6, 3))))
    return out


# (GE2E_FUZZ_SEED / GE2E_FUZZ_N: a longer one-off soak with other shapes; the committed default is what the suite runs)
@pytest.mark.parametrize("case", cases(int(os.environ.get("GE2E_FUZZ_N", "48")), seed=int(os.environ.get("GE2E_FUZZ_SEED", "2024"))), ids=lambda c: f"B{c[0]}_N{c[1]}_M{c[2]}_D{c[3]}_{c[4][0]}")
def test_random_shapes(GF, case):
    B, N, M, D, variant, w, b = case
    E = orc.synth_embeddings((B, N, M, D), "raw" if (N + M) % 3 == 0 else "unit", seed=B * 7 + N)
    ref = orc.closed_form(E, w, b, variant=variant)
    if variant == "contrast":
        # eq. 7's max over the other speakers is not smooth: where the two largest similarities of a row are closer than
        # fp32 resolves, fp32 and fp64 may pick different speakers and the row's gradient moves to another centroid
        S = w * np.asarray(ref["cos"], np.float64) + b
        jj = np.arange(N)
        S[:, jj, :, jj] = -np.inf
        top2 = np.sort(S, axis=-1)[..., -2:]
        if N > 2 and float((top2[..., 1] - top2[..., 0]).min()) < 5e-6 * max(1.0, float(np.abs(top2[..., 1]).max())):
            pytest.skip("contrast: a row's two largest similarities tie within fp32 resolution")
    # AUTO keeps the matrix-core kernels for every row-aligned D: no shape with N >= 13 may land on the VALU fall-back (85x
    # slower than its neighbours) when D % 4 == 0 (N <= 64, D <= 256: padded inside the team / one-workgroup kernels) or
    # D % 8 == 0 (the tiled pipeline: rows of its fp16 planes 16-byte aligned).  Left to the fall-back: D % 4 != 0 (rows of E
    # not 16-byte aligned), and D % 8 == 4 with N > 64 or D > 256.
    if N >= 13 and ((N <= 64 and D <= 256 and D % 4 == 0) or (D <= 1024 and D % 8 == 0)):
        assert GF.resolve_impl(B, N, M, D, variant, "auto") != "generic", case
    seen = set()
    for impl in impls_for(GF, B, N, M, D, variant):
        resolved = GF.resolve_impl(B, N, M, D, variant, impl)
        if resolved in seen and impl == "auto":
            continue
        seen.add(resolved)
        check(run_hip(GF, E, w, b, variant, impl), ref, impl, f"{case}/{impl}->{resolved}")
