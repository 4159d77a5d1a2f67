"""Batch sampler (s1_dataset_loader.py:52-77) against items returned by the reference's own dataset class
(tests/golden/make_sampler_golden.py).  CPU: the host draws (file shuffle, utterance indices, crop start) reproduce the
reference's under the same seeds.  GPU: the gather-and-cast kernel returns the same batch, bit for bit (as float32)."""
import os
import random

import numpy as np
import pytest
import torch

from speaker_embedding_ge2e_loss_amd.data import GE2EBatchSampler, SpectrogramStore

Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "callers", "sampler.npz"))
NAMES = [str(n) for n in Z["walk_order"]]
ARRAYS = [Z["file." + n] for n in NAMES]


class HostStore:
    """What the sampler's host logic reads of a store (no device)."""
    def __init__(self):
        self.names, self.utterances, self.T, self.F = NAMES, [a.shape[0] for a in ARRAYS], int(Z["T"]), int(Z["F"])

    def __len__(self):
        return len(self.names)


@pytest.mark.parametrize("mode", ["train", "test"])
def test_host_draws_are_the_references(mode):
    M, L, seed = [int(v) for v in Z[mode + ".cfg"]]
    random.seed(seed)
    s = GE2EBatchSampler(HostStore(), M, L, training=(mode == "train"))
    assert [NAMES[i] for i in s.order] == [str(n) for n in Z[mode + ".order"]]   # s1:40: shuffled only when training
    np.random.seed(seed + 1)
    for idx in range(len(s)):
        spk, utt, clip = s.draw(idx)
        want = Z[mode + ".items"][idx]
        got = ARRAYS[spk][utt, clip:clip + L, :]       # s1:68, 74 -- test-side numpy, the product gathers on the GPU
        assert np.array_equal(got, want), (mode, idx)


def test_crop_length_must_leave_room():
    with pytest.raises(ValueError):
        GE2EBatchSampler(HostStore(), 3, int(Z["T"]) - 1)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["train", "test"])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_gpu_batch_equals_reference_items(mode, dtype):
    M, L, seed = [int(v) for v in Z[mode + ".cfg"]]
    store = SpectrogramStore([a.astype(dtype) for a in ARRAYS], NAMES, "cuda:0")
    random.seed(seed)
    s = GE2EBatchSampler(store, M, L, training=(mode == "train"))
    np.random.seed(seed + 1)
    out = s.batch(range(len(s)))                       # one "DataLoader batch" of all six speakers, in dataset order
    assert out.shape == (6, M, L, int(Z["F"])) and out.dtype == torch.float32 and out.is_cuda
    want = Z[mode + ".items"].astype(np.float32)       # the cast the encoder makes at s2:28
    assert np.array_equal(out.cpu().numpy(), want)


@pytest.mark.gpu
def test_loader_batches_and_feeds_the_trainer_shape():
    store = SpectrogramStore(ARRAYS, NAMES, "cuda:0")
    s = GE2EBatchSampler(store, 4, 12, training=False)
    batches = list(s.loader(batch_size=4, shuffle=True, generator=torch.Generator().manual_seed(1)))
    assert len(batches) == 1 and batches[0].shape == (4, 4, 12, int(Z["F"]))      # 6 speakers, drop_last
    assert len(list(s.loader(batch_size=2, shuffle=False))) == 3
    assert torch.isfinite(batches[0]).all()
    with pytest.raises(RuntimeError):
        SpectrogramStore(ARRAYS, NAMES, "cpu")
