"""Equal-error-rate fixtures from the reference's own ``calculate_ERR`` (build container only).

    python tests/golden/make_eer_golden.py

``embedding_model_GE2E/s5_eval_model.py:16-100`` is run AS IT IS: the module is imported from
/root/reference, its data-loader factory (s5:21; reads spectrogram folders that do not exist here)
is replaced by one that yields our synthetic batch, and the "model" handed in returns precomputed
embeddings.  The function only PRINTS its result (two decimals, s5:100), so the script
 * parses that line into ``printed`` = (EER, thres, FAR, FRR), and
 * recomputes the sweep at full precision with the reference's ``GE2ELoss.get_cos_sim`` output and
   the statement sequence of s5:57-98, checks that it rounds to the printed line, and stores the
   integer counts per threshold plus the unrounded result.
Data only: embeddings in, similarity matrix / counts / result out.
"""
import contextlib
import io
import os
import re
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")

import embedding_model_GE2E.s5_eval_model as s5  # noqa: E402
from embedding_model_GE2E.s3_loss_function_GE2E import GE2ELoss  # noqa: E402
from utils.dict_to_dot import GetDictWithDotNotation  # noqa: E402


def speakers(N, M, D, sigma, seed, outliers=()):
    """Unit speaker directions + isotropic noise of expected norm sigma; `outliers` (j, i) get 5x the noise."""
    g = torch.Generator().manual_seed(seed)
    c = torch.nn.functional.normalize(torch.randn(N, 1, D, generator=g), dim=-1)
    noise = sigma / D ** 0.5 * torch.randn(N, M, D, generator=g)
    for j, i in outliers:
        noise[j, i] *= 5.0
    return torch.nn.functional.normalize(c + noise, dim=-1)


def run_reference(emb):
    N, M, D = emb.shape
    hp = GetDictWithDotNotation({"general": {"small_err": 1e-6, "device": torch.device("cpu")},
                                 "m_ge2e": {"test_N": N, "test_M": M}})
    mel = emb.reshape(1, N * M, 1, D)  # (batch, N*M, frames=1, "mels"=D): reshaped to (N*M, 1, D) at s5:32-33
    s5.get_train_test_data_loader = lambda hp: (None, [mel])
    model = lambda x: x[:, 0, :]  # noqa: E731  the "encoder": hands the stored embeddings back
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        s5.calculate_ERR(model, hp, N=N, M=M)
    m = re.search(r"EER : ([-\d.]+) \(thres:([-\d.]+), FAR:([-\d.]+), FRR:([-\d.]+)\)", buf.getvalue())
    printed = np.array([float(x) for x in m.groups()])

    # full precision, same statements (s5:42-98)
    cent = GE2ELoss.get_centroids(emb)
    cos = GE2ELoss.get_cos_sim(emb, cent, hp)
    S = (torch.tensor(1.0) * cos + torch.tensor(0.0)).detach().cpu().numpy()
    diff, EER, EER_thres, EER_FAR, EER_FRR = 1, 0, 0, 0, 0
    thres_lst = [0.01 * i + 0.5 for i in range(50)]
    counts = []
    for thres in thres_lst:
        S_thres = S > thres
        fa = sum([np.sum(S_thres[i]) - np.sum(S_thres[i, :, i]) for i in range(N)])
        ta = sum([np.sum(S_thres[i][:, i]) for i in range(N)])
        counts.append((int(fa), int(ta)))
        FAR = fa / ((N - 1) / M / N)
        FRR = sum([M - np.sum(S_thres[i][:, i]) for i in range(N)]) / (M / N)
        if diff > abs(FAR - FRR):
            diff = abs(FAR - FRR)
            EER, EER_thres, EER_FAR, EER_FRR = (FAR + FRR) / 2, thres, FAR, FRR
    full = np.array([EER, EER_thres, EER_FAR, EER_FRR], dtype=np.float64)
    assert np.allclose(np.round(full, 2), printed, atol=0.0051), (full, printed)
    return S, np.array(counts, dtype=np.int64), full, printed


def main():
    cases = {
        # name: (N, M, D, sigma, seed, outliers)
        "eer_separated": (4, 16, 64, 0.3, 1, ()),          # FAR = FRR = 0 at the first threshold
        "eer_default": (4, 16, 256, 0.6, 2, ((1, 3), (2, 7))),  # s5's default N, M; two rejected utterances
        "eer_noisy": (4, 16, 32, 0.9, 3, ()),
        "eer_never": (4, 16, 16, 5.0, 4, ()),              # |FAR - FRR| never < 1: the initial zeros survive (s5:50-54)
        "eer_n8_m10": (8, 10, 128, 0.5, 5, ((6, 2),)),
        "eer_n64_m10": (64, 10, 256, 0.4, 6, ()),          # the metric shape
    }
    out = {}
    for name, (N, M, D, sigma, seed, outl) in cases.items():
        emb = speakers(N, M, D, sigma, seed, outl)
        S, counts, full, printed = run_reference(emb)
        out[name + ".E"] = emb.numpy()
        out[name + ".S"] = S
        out[name + ".counts"] = counts
        out[name + ".result"] = full
        out[name + ".printed"] = printed
        print(f"{name:14s} N={N} M={M} D={D}: EER {full[0]:.4f} thres {full[1]:.2f} FAR {full[2]:.4f} FRR {full[3]:.4f}"
              f"  counts[0]={counts[0].tolist()} counts[-1]={counts[-1].tolist()}")
    path = os.path.join(HERE, "callers", "eer.npz")
    np.savez_compressed(path, **out)
    print(f"{path}: {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
