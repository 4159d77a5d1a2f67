"""calculate_ERR (s5:16-100) against fixtures printed and recomputed by the reference's own function
(tests/golden/make_eer_golden.py).  CPU: the scalar sweep on the stored counts.  GPU: cosines + counts
through the C ABI, bit-exact integers, then the same result."""
import os

import numpy as np
import pytest
import torch

from speaker_embedding_ge2e_loss_amd import evaluation as EV

Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "callers", "eer.npz"))
CASES = sorted({k.split(".")[0] for k in Z.files})


@pytest.mark.parametrize("name", CASES)
def test_sweep_on_reference_counts(name):
    N, M, _ = Z[name + ".E"].shape
    r = EV.eer_from_counts(Z[name + ".counts"], N, M)
    got = np.array([r["EER"], r["thres"], r["FAR"], r["FRR"]], dtype=np.float64)
    assert np.array_equal(got, Z[name + ".result"]), (got, Z[name + ".result"])
    # what the reference prints (two decimals, s5:100)
    assert np.allclose(np.round(got, 2), Z[name + ".printed"], atol=0.0051)


def test_the_references_denominators_are_kept():
    """s5:81,88: FAR = fa / ((N-1)/M/N), FRR = rejected / (M/N) -- not ratios.  One reject at N=8, M=10 is
    FRR 0.8 and EER 0.4 in the reference; the textbook rate would be 1/80."""
    counts = np.array([[0, 79]] + [[0, 0]] * 49)
    r = EV.eer_from_counts(counts, 8, 10)
    assert r == {"EER": 0.4, "thres": 0.5, "FAR": 0.0, "FRR": 0.8}
    rn = EV.eer_from_counts(counts, 8, 10, normalized=True)
    assert rn["FRR"] == 1 / 80 and rn["FAR"] == 0.0
    # nothing within |FAR - FRR| < 1: the initial zeros survive, threshold included (s5:50-54)
    assert EV.eer_from_counts(np.array([[7, 2]] * 50), 4, 16) == {"EER": 0, "thres": 0, "FAR": 0, "FRR": 0}


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_counts_and_result_on_gpu(name):
    from speaker_embedding_ge2e_loss_amd import functional as GF
    dev = torch.device("cuda:0")
    S = torch.from_numpy(Z[name + ".S"]).to(dev)
    counts = GF.eer_counts(S, EV.THRESHOLDS).cpu().numpy()
    assert counts.dtype == np.int32 and np.array_equal(counts, Z[name + ".counts"])  # integers: exact
    r = EV.eer_from_sim(S)
    assert np.array_equal(np.array([r["EER"], r["thres"], r["FAR"], r["FRR"]]), Z[name + ".result"])


@pytest.mark.gpu
def test_calculate_err_end_to_end_on_gpu(capsys):
    """The reference's call shape: model, hp, N, M (+ the loader it would have built).  Embeddings -> HIP
    cosines -> HIP counts.  A cosine within one fp32 ulp of a threshold could flip a count against the
    reference's torch cosines, so the end-to-end result is compared on fixtures whose cosines keep clear of
    the grid (checked here), and batched input is checked against the per-batch calls."""
    from speaker_embedding_ge2e_loss_amd import HParams, functional as GF
    dev = torch.device("cuda:0")
    hp = HParams(device=dev)
    for name in CASES:
        E = torch.from_numpy(Z[name + ".E"])
        N, M, D = E.shape
        thr = np.array(EV.THRESHOLDS, dtype=np.float32)
        margin = np.abs(Z[name + ".S"][..., None] - thr).min()
        res = EV.calculate_ERR(lambda x: x[:, 0, :], hp, N=N, M=M, test_loader=[E.reshape(1, N * M, 1, D)])
        assert hp.m_ge2e.test_N == N and hp.m_ge2e.test_M == M
        line = capsys.readouterr().out
        assert "EER : %0.2f (thres:%0.2f, FAR:%0.2f, FRR:%0.2f)" % tuple(Z[name + ".result"]) in line or margin < 2e-6
        if margin >= 2e-6:
            got = np.array([res[0]["EER"], res[0]["thres"], res[0]["FAR"], res[0]["FRR"]])
            assert np.array_equal(got, Z[name + ".result"]), name
    # batched: three (4,16,*) similarity matrices in one launch
    names = ["eer_separated", "eer_noisy", "eer_never"]
    Sb = torch.stack([torch.from_numpy(Z[n + ".S"]) for n in names]).to(dev)
    cb = GF.eer_counts(Sb, EV.THRESHOLDS).cpu().numpy()
    for i, n in enumerate(names):
        assert np.array_equal(cb[i], Z[n + ".counts"])
    with pytest.raises(ValueError):
        GF.eer_counts(Sb, [0.6, 0.5])


@pytest.mark.gpu
def test_eer_counts_ties_nan_and_random_tables():
    """`>` is strict (a value equal to a threshold is not accepted), NaN accepts nothing, repeated thresholds
    allowed; checked against numpy's fp32 comparison on random data at an odd shape."""
    from speaker_embedding_ge2e_loss_amd import functional as GF
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(5)
    N, M, B, T = 7, 3, 5, 33
    S = rng.uniform(0.3, 1.0, size=(B, N, M, N)).astype(np.float32)
    thr = np.sort(rng.uniform(0.3, 1.0, size=T)).astype(np.float32)
    thr[5] = thr[4]
    S[0, 1, 2, 3] = thr[10]
    S[1, 2, 0, 2] = thr[20]
    S[2, 0, 0, 0] = np.nan
    S[3, 6, 2, 1] = np.inf
    got = GF.eer_counts(torch.from_numpy(S).to(dev), thr.astype(np.float64)).cpu().numpy()
    own = np.eye(N, dtype=bool)[:, None, :]
    for b in range(B):
        for t in range(T):
            acc = S[b] > thr[t]
            assert got[b, t, 0] == int((acc & ~own).sum()) and got[b, t, 1] == int((acc & own).sum()), (b, t)


def test_entry_points_have_the_reference_signatures():
    """s5:16 `calculate_ERR(model, hp, N=4, M=16)` and s4:19 / s4:137 `TrainEmbedModel(hp).train_model(lr_reduce=2000,
    epoch_print=100, dot_print=10)`: positional order and defaults as in the reference, so its scripts can switch imports."""
    import inspect
    from speaker_embedding_ge2e_loss_amd.evaluation import calculate_ERR
    from speaker_embedding_ge2e_loss_amd.trainer import TrainEmbedModel
    from speaker_embedding_ge2e_loss_amd.data import get_train_test_data_loader
    ps = list(inspect.signature(calculate_ERR).parameters.values())
    assert [p.name for p in ps[:4]] == ["model", "hp", "N", "M"] and ps[2].default == 4 and ps[3].default == 16
    assert all(p.default is not inspect.Parameter.empty for p in ps[2:])
    pi = list(inspect.signature(TrainEmbedModel.__init__).parameters.values())
    assert [p.name for p in pi[:2]] == ["self", "hp"] and all(p.default is not inspect.Parameter.empty for p in pi[2:])
    pt = inspect.signature(TrainEmbedModel.train_model).parameters
    assert [(n, p.default) for n, p in list(pt.items())[1:]] == [("lr_reduce", 2000), ("epoch_print", 100), ("dot_print", 10)]
    assert list(inspect.signature(get_train_test_data_loader).parameters) == ["hp"]
