"""Equal-error-rate evaluation: the caller on the output side of the hot path.

Mirrors ``embedding_model_GE2E/s5_eval_model.py:16-100`` (``calculate_ERR``): per test batch the
encoder's embeddings -> get_centroids -> get_cos_sim -> sim = w cos + b with w = 1, b = 0 (s5:27-28,
44-46), then a sweep over the 50 thresholds 0.5 + 0.01 i (s5:57) that keeps the threshold where
|FAR - FRR| is smallest.  The cosines and the per-threshold counts run in libge2e_hip.so
(``ge2e_cos_sim``, ``ge2e_eer_counts``); only 100 integers per batch come back to the host, where
the reference's own scalar arithmetic is repeated verbatim -- including its denominators
(s5:81: (N-1)/M/N for FAR, s5:88: M/N for FRR), which are NOT the population sizes the comments
describe, so FAR/FRR are not ratios in [0,1]; a drop-in has to reproduce them, not repair them.
``normalized=True`` gives the textbook rates instead.
"""
from __future__ import annotations

import torch

from . import functional as GF

THRESHOLDS = [0.01 * i + 0.5 for i in range(50)]  # s5:57


def eer_from_counts(counts, N: int, M: int, thresholds=THRESHOLDS, normalized: bool = False):
    """The scalar part of s5:50-98 for one batch.  counts: (T,2) integers [false accepts, own accepts]."""
    diff, EER, EER_thres, EER_FAR, EER_FRR = 1, 0, 0, 0, 0  # s5:50-54
    for thres, (fa, ta) in zip(thresholds, counts):
        fa, ta = int(fa), int(ta)
        if normalized:
            FAR = fa / ((N - 1) * M * N)
            FRR = (N * M - ta) / (N * M)
        else:
            FAR = fa / ((N - 1) / M / N)      # s5:81-83
            FRR = (N * M - ta) / (M / N)      # s5:88-90: sum_i (M - accepted_i)
        if diff > abs(FAR - FRR):             # s5:93-98
            diff = abs(FAR - FRR)
            EER = (FAR + FRR) / 2
            EER_thres = thres
            EER_FAR = FAR
            EER_FRR = FRR
    return {"EER": EER, "thres": EER_thres, "FAR": EER_FAR, "FRR": EER_FRR}


def eer_from_sim(sim_matrix: torch.Tensor, thresholds=THRESHOLDS, normalized: bool = False):
    """(N,M,N) -> dict, (B,N,M,N) -> list of dicts.  One kernel, one (T,2)-per-batch readback."""
    counts = GF.eer_counts(sim_matrix, thresholds).cpu().numpy()
    N, M = sim_matrix.shape[-3], sim_matrix.shape[-2]
    if sim_matrix.dim() == 3:
        return eer_from_counts(counts, N, M, thresholds, normalized)
    return [eer_from_counts(c, N, M, thresholds, normalized) for c in counts]


def calculate_ERR(model, hp, N: int = 4, M: int = 16, test_loader=None, verbose: bool = True):
    """s5:16-100, the reference's signature: ``calculate_ERR(model, hp, N, M)`` writes N, M into ``hp.m_ge2e.test_N`` /
    ``test_M`` (s5:17-18) and builds the test loader from ``hp`` (s5:21) -- here ``data.get_train_test_data_loader``'s
    second result: the ``sv_*.npy`` folder ``hp.m_ge2e.tt_data.test_spects_path`` resident in HBM, batches gathered by one
    kernel.  ``test_loader`` (an iterable of (N,M,T,F) mel batches) replaces that default for callers that have their own.
    Prints the reference's result line per batch and returns the results (the reference returns None)."""
    hp.m_ge2e.test_N, hp.m_ge2e.test_M = N, M  # s5:17-18
    if test_loader is None:
        from .data import get_train_test_data_loader
        _, test_loader = get_train_test_data_loader(hp)   # s5:21
    total = N * M
    results = []
    with torch.no_grad():
        for mel in test_loader:
            mel = torch.reshape(mel, (total, mel.size(2), mel.size(3))).to(hp.general.device)  # s5:32-33
            emb = model(mel)                                                                   # s5:36
            emb = torch.reshape(emb, (N, M, emb.size(1)))                                      # s5:39
            cos = GF.cos_sim(emb, eps=hp.general.small_err)                                    # s5:42-43
            sim = 1.0 * cos + 0.0                                                              # s5:44 (w=1, b=0)
            r = eer_from_sim(sim)
            if verbose:
                print("\nEER : %0.2f (thres:%0.2f, FAR:%0.2f, FRR:%0.2f)" % (r["EER"], r["thres"], r["FAR"], r["FRR"]))
            results.append(r)
    return results
