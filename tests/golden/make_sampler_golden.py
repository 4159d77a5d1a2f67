"""Sampler fixtures from the reference's own dataset class (build container only).

    python tests/golden/make_sampler_golden.py

Writes a few synthetic ``sv_*.npy`` speaker files ((U, T, F) float64, the reference's on-disk format) to a temporary
folder, instantiates ``embedding_model_GE2E/s1_dataset_loader.py:EmbeddingModelTTDataset`` on it (training and test
mode), seeds ``random`` / ``np.random`` and calls ``__getitem__`` for every speaker in order -- what the DataLoader does
for one batch.  Stored: the speaker arrays, the file order after the dataset's shuffle, the seeds, and the float64 items
it returned.  (``get_train_test_data_loader`` itself cannot run on this torch: it passes ``prefetch_factor`` without
workers, s1:97.)  Data only.
"""
import os
import random
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")

from embedding_model_GE2E.s1_dataset_loader import EmbeddingModelTTDataset  # noqa: E402
from utils.dict_to_dot import GetDictWithDotNotation  # noqa: E402


def main():
    rng = np.random.default_rng(7)
    T, F = 30, 6
    out = {"T": np.int64(T), "F": np.int64(F)}
    with tempfile.TemporaryDirectory() as d:
        for j, U in enumerate([3, 7, 4, 5, 9, 2]):
            a = rng.standard_normal((U, T, F)) * 3.0 + j  # float64, values that do not survive a float32 round trip unchanged
            np.save(os.path.join(d, f"sv_spk{j}.npy"), a)
            out[f"file.sv_spk{j}.npy"] = a
        for mode, training, M, L, seed in (("train", True, 5, 20, 123), ("test", False, 4, 16, 321)):
            hp = GetDictWithDotNotation({"m_ge2e": {"training_M": M, "test_M": M,
                                                    "tt_data": {"min_train_utter_len": L, "min_test_utter_len": L}}})
            random.seed(seed)
            ds = EmbeddingModelTTDataset(d, hp, training=training)
            np.random.seed(seed + 1)
            items = np.stack([ds[i] for i in range(len(ds))])  # (speakers, M, L, F) float64
            assert items.dtype == np.float64 and items.shape == (6, M, L, F)
            out[f"{mode}.order"] = np.array([os.path.basename(f) for f in ds.lst_spkr_np_files])
            out[f"{mode}.cfg"] = np.array([M, L, seed], dtype=np.int64)
            out[f"{mode}.items"] = items
            print(mode, [str(f) for f in out[f"{mode}.order"]], items.shape, float(items.mean()))
        out["walk_order"] = np.array(next(iter(os.walk(d)))[2])
    path = os.path.join(HERE, "callers", "sampler.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
