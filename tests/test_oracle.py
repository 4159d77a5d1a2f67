"""CPU: pin both oracle restatements against the reference-generated golden vectors."""
import numpy as np
import pytest

from conftest import rel_fro
from oracle import ge2e_oracle as orc


def test_closed_form_fp64_matches_reference_fp64(golden):
    g = golden
    o = orc.closed_form(g["E"], float(g["w"]), float(g["b"]), dtype=np.float64)
    assert np.allclose(o["loss"], g["loss64"], rtol=1e-12, atol=1e-12)
    assert np.allclose(o["per"], g["per64"], rtol=1e-10, atol=1e-12)
    assert np.allclose(o["cos"], g["cos64"], rtol=1e-12, atol=1e-14)
    assert np.allclose(o["dw"], g["dw64"], rtol=1e-10, atol=1e-12)
    assert np.allclose(o["db"], g["db64"], rtol=1e-7, atol=1e-13)
    if "dE64" in g:
        assert rel_fro(o["dE"], g["dE64"]) < 1e-12
        assert np.abs(o["dE"] - g["dE64"]).max() <= 1e-12 * max(1.0, np.abs(g["dE64"]).max())


def test_closed_form_fp32_within_reference_fp32_noise(golden):
    """The fp32 closed form and the fp32 reference both sit ~1e-7 from the fp64 truth."""
    g = golden
    o = orc.closed_form(g["E"], float(g["w"]), float(g["b"]), dtype=np.float32)
    assert np.allclose(o["loss"], g["loss64"], rtol=2e-6, atol=1e-5)
    assert np.allclose(o["dw"], g["dw64"], rtol=2e-5, atol=1e-5)
    assert np.allclose(o["db"], g["db64"], atol=1e-4)
    assert np.allclose(o["cos"], g["cos64"], atol=2e-6)


def test_expand_form_reproduces_reference_fp32(golden):
    """Same ATen op sequence as s3 -> matches the reference's fp32 numbers to the last bits."""
    g = golden
    o = orc.expand_form_loss_and_grads(g["E"], float(g["w"]), float(g["b"]))
    assert np.allclose(o["loss"], g["loss"], rtol=1e-6, atol=1e-6)
    assert np.allclose(o["per"], g["per"], rtol=1e-5, atol=1e-6)
    assert np.allclose(o["cos"], g["cos"], rtol=0, atol=1e-6)
    assert np.allclose(o["dw"], g["dw"], rtol=1e-5, atol=1e-5)
    assert np.allclose(o["db"], g["db"], atol=1e-4)
    # degenerate rows have 1e8-scale gradients: compare relative to the largest entry
    assert np.abs(o["dE"] - g["dE"]).max() <= 1e-5 * max(1.0, np.abs(g["dE"]).max())
    assert rel_fro(o["dE"], g["dE"]) < 1e-5


def test_expand_form_fp64_matches_reference_fp64(golden):
    import torch
    g = golden
    o = orc.expand_form_loss_and_grads(g["E"], float(g["w"]), float(g["b"]), dtype=torch.float64)
    assert np.allclose(o["loss"], g["loss64"], rtol=1e-13)
    assert np.allclose(o["dw"], g["dw64"], rtol=1e-11, atol=1e-13)
    if "dE64" in g:
        assert rel_fro(o["dE"], g["dE64"]) < 1e-13


def test_known_answer_toy_values():
    """SURVEY.md section 4 / s3:144-150: the one known-answer input in the reference."""
    toy = np.array([[0, 1, 0], [0, 0, 1], [0, 1, 0], [0, 1, 0], [1, 0, 0], [1, 0, 0]],
                   dtype=np.float32).reshape(3, 2, 3)
    o = orc.closed_form(toy, 1.0, 0.0)
    assert abs(o["loss"] - 5.25009441) < 1e-6
    ref_per = [[1.55144501, 1.09861267], [0.74857318, 0.74857318], [0.55144501, 0.55144501]]
    assert np.allclose(o["per"], ref_per, atol=1e-6)
    o = orc.closed_form(toy, 10.0, -5.0)
    assert abs(o["loss"] - 11.20316792) < 1e-5
    assert abs(o["dw"] - 0.96991879) < 1e-6
    # db is a cancellation residue (SURVEY 7 'Tolerance'): the survey's fp32 figure is
    # -4.9572e-05, the fp64 truth -4.9501e-05 -> atol only
    assert abs(o["db"] - (-4.9572e-05)) < 1e-6


@pytest.mark.parametrize("kind", ["unit", "raw", "clustered"])
@pytest.mark.parametrize("shape", [(4, 5, 256), (7, 3, 40), (16, 2, 64)])
def test_contrast_two_forms_agree(kind, shape):
    """Contrast variant (paper eq. 7) is not in the reference: parity unpinned.
    The two independent restatements (autograd vs hand-derived) must agree."""
    import torch
    e = orc.synth_embeddings(shape, kind, seed=11)
    a = orc.expand_form_loss_and_grads(e, 10.0, -5.0, variant="contrast", dtype=torch.float64)
    c = orc.closed_form(e, 10.0, -5.0, variant="contrast")
    assert np.allclose(a["loss"], c["loss"], rtol=1e-12)
    assert np.allclose(a["per"], c["per"], rtol=1e-10, atol=1e-12)
    assert np.allclose(a["dw"], c["dw"], rtol=1e-9, atol=1e-12)
    assert np.allclose(a["db"], c["db"], rtol=1e-9, atol=1e-12)
    assert rel_fro(c["dE"], a["dE"]) < 1e-11


def test_closed_form_gradient_by_finite_differences():
    """Independent of autograd: central differences on the fp64 closed form."""
    e = orc.synth_embeddings((3, 3, 6), "raw", seed=3).astype(np.float64)
    for variant in ("softmax", "contrast"):
        o = orc.closed_form(e, 4.0, -1.0, variant=variant)
        h = 1e-6
        num = np.zeros_like(e)
        for idx in np.ndindex(e.shape):
            ep = e.copy(); ep[idx] += h
            em = e.copy(); em[idx] -= h
            num[idx] = (orc.closed_form(ep, 4.0, -1.0, variant=variant, want_grad=False)["loss"]
                        - orc.closed_form(em, 4.0, -1.0, variant=variant, want_grad=False)["loss"]) / (2 * h)
        assert np.allclose(num, o["dE"], rtol=1e-5, atol=1e-7)
        dw = (orc.closed_form(e, 4.0 + h, -1.0, variant=variant, want_grad=False)["loss"]
              - orc.closed_form(e, 4.0 - h, -1.0, variant=variant, want_grad=False)["loss"]) / (2 * h)
        assert abs(dw - o["dw"]) < 1e-6


def test_reference_quirks_are_kept():
    e = orc.synth_embeddings((4, 3, 16), "unit", seed=2)
    # w is not clamped (s3:22 no-op): negative w changes the loss
    assert orc.closed_form(e, -3.0, 0.0)["loss"] != orc.closed_form(e, 1e-6, 0.0)["loss"]
    # sum, not mean (s3:126)
    o = orc.closed_form(e, 10.0, -5.0)
    assert np.isclose(o["loss"], o["per"].sum())
    # unstabilised exp overflows at huge w like the reference; the stable form stays finite
    assert np.isinf(orc.closed_form(e, 1e5, 0.0, stable=False)["loss"])
    assert np.isfinite(orc.closed_form(e, 1e5, 0.0, stable=True)["loss"])
    # batched input
    eb = orc.synth_embeddings((3, 4, 3, 16), "unit", seed=5)
    ob = orc.closed_form(eb)
    assert ob["dE"].shape == eb.shape and ob["loss"].shape == (3,)
    assert np.isclose(ob["loss"][1], orc.closed_form(eb[1])["loss"])
