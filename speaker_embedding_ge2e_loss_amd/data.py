"""GPU-side batch sampler: the caller that FEEDS the training step.

Mirrors ``embedding_model_GE2E/s1_dataset_loader.py``: ``EmbeddingModelTTDataset`` keeps one
``(U, T, F)`` float64 array per speaker on disk (``sv_<speaker>.npy``), and per item (s1:52-77) loads
the file, draws ``M`` utterance indices with replacement (s1:65), one crop start (s1:71), slices and
returns float64; the DataLoader stacks N speakers (s1:93-104) and the encoder casts to float32 (s2:28).

Here the arrays are loaded ONCE and stay resident in HBM in their on-disk dtype (1.1 GB for 100 speakers x
100 utterances at 180 x 80 float64 -- nothing beside 288 GB); a batch is the same host draws (same
``np.random`` calls in the same order, so a seeded run picks the reference's utterances and crops) + ONE
gather-and-cast launch (``ge2e_sample_batch``): no per-item file read, no host slicing, no f64 -> f32
host cast, no H2D copy of the batch -- only N*M + N int32 indices cross PCIe.
"""
from __future__ import annotations

import os
import random
from typing import Iterator, List, Optional, Sequence

import numpy as np
import torch

from . import _lib


class SpectrogramStore:
    """Every speaker's (U_j, T, F) array in one device buffer; ``offsets`` in elements."""

    def __init__(self, arrays: Sequence[np.ndarray], names: Optional[Sequence[str]] = None, device="cuda:0"):
        if not arrays:
            raise ValueError("no speaker arrays")
        T, F, dt = arrays[0].shape[1], arrays[0].shape[2], arrays[0].dtype
        if dt not in (np.float64, np.float32):
            raise TypeError(f"spectrograms must be float64 (the reference's on-disk type) or float32, got {dt}")
        for a in arrays:
            if a.ndim != 3 or a.shape[1] != T or a.shape[2] != F or a.dtype != dt:
                raise ValueError("every speaker array must be (U, T, F) with the same T, F and dtype")
        self.T, self.F, self.dtype = T, F, dt
        self.names = list(names) if names is not None else [str(i) for i in range(len(arrays))]
        self.utterances = [int(a.shape[0]) for a in arrays]
        sizes = [int(a.size) for a in arrays]
        self.offsets = [0]
        for s in sizes[:-1]:
            self.offsets.append(self.offsets[-1] + s)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("SpectrogramStore lives on the GPU (there is no CPU fallback)")
        flat = torch.from_numpy(np.concatenate([np.ascontiguousarray(a).reshape(-1) for a in arrays]))
        self.buffer = flat.pin_memory().to(self.device, non_blocking=True)

    @classmethod
    def from_dir(cls, data_path: str, device="cuda:0") -> "SpectrogramStore":
        files = next(iter(os.walk(data_path)))[2]  # s1:21-30: the files of the folder, in os.walk's order
        return cls([np.load(os.path.join(data_path, f)) for f in files], files, device)

    def __len__(self):
        return len(self.utterances)


class GE2EBatchSampler:
    """``EmbeddingModelTTDataset`` + the stacking of its DataLoader, for a resident store."""

    def __init__(self, store: SpectrogramStore, utter_num: int, min_utter_len: int, training: bool = True):
        self.store, self.utter_num, self.min_utter_len, self.training = store, int(utter_num), int(min_utter_len), training
        if not 0 < self.min_utter_len < store.T - 1:
            raise ValueError("min_utter_len must leave room for s1:71's randint(0, T - min_utter_len - 1)")
        self.order = list(range(len(store)))
        if training:
            random.shuffle(self.order)  # s1:40 shuffles the file list with the global python RNG

    @classmethod
    def from_hp(cls, store, hp, training=True):
        m = hp.m_ge2e  # s1:36-46
        return cls(store, m.training_M if training else m.test_M,
                   m.tt_data.min_train_utter_len if training else m.tt_data.min_test_utter_len, training)

    def __len__(self):
        return len(self.store)

    def draw(self, idx: int):
        """The two draws of ``__getitem__`` (s1:65, s1:71), same calls on the same global numpy RNG."""
        spk = self.order[idx]
        utter_idx = np.random.randint(0, self.store.utterances[spk], self.utter_num)
        clip = np.random.randint(0, self.store.T - self.min_utter_len - 1)
        return spk, utter_idx, int(clip)

    def batch(self, indices: Sequence[int]) -> torch.Tensor:
        """(N, M, min_utter_len, F) float32 on the device for dataset indices ``indices`` (one DataLoader batch)."""
        st = self.store
        N, M, L = len(indices), self.utter_num, self.min_utter_len
        off = np.empty(N, dtype=np.int64)
        utt = np.empty((N, M), dtype=np.int32)
        clip = np.empty(N, dtype=np.int32)
        for n, idx in enumerate(indices):  # the DataLoader calls __getitem__ in batch order
            spk, u, c = self.draw(idx)
            off[n], utt[n], clip[n] = st.offsets[spk], u, c
        meta = np.concatenate([off.view(np.int32), utt.reshape(-1), clip])  # one pinned staging buffer, one async copy
        dmeta = torch.from_numpy(meta).pin_memory().to(st.device, non_blocking=True)
        out = torch.empty(N, M, L, st.F, dtype=torch.float32, device=st.device)
        base = dmeta.data_ptr()
        with torch.cuda.device(st.device):
            code = _lib.load().ge2e_sample_batch(
                st.buffer.data_ptr(), 1 if st.dtype == np.float64 else 0, base, base + 8 * N, base + 8 * N + 4 * N * M,
                N, M, st.T, L, st.F, out.data_ptr(), torch.cuda.current_stream(st.device).cuda_stream)
        _lib.check(code, "ge2e_sample_batch")
        # dmeta may go out of scope here: torch's allocator frees stream-ordered, and the launch above is already
        # enqueued on the stream any re-use of the block would be ordered behind
        return out

    def loader(self, batch_size: int, shuffle: bool = True, generator: Optional[torch.Generator] = None) -> "SpeakerBatchLoader":
        """Batches of ``batch_size`` speakers, ``drop_last=True`` like the DataLoader of s1:93-104.  The result is
        RE-ITERABLE: every ``iter()`` (every epoch of ``DPTrainer.fit``) draws a fresh speaker order, as a DataLoader over
        a RandomSampler does.  The order is one ``torch.randperm`` from ``generator`` (or torch's global generator) per
        epoch -- the same distribution as the reference's RandomSampler, not the same sequence: torch's sampler first draws
        a seed from the global generator and permutes with a generator of its own."""
        return SpeakerBatchLoader(self, batch_size, shuffle, generator)


class SpeakerBatchLoader:
    """Re-iterable view of a GE2EBatchSampler: one pass = one epoch of (N,M,L,F) device batches."""

    def __init__(self, sampler, batch_size: int, shuffle: bool, generator: Optional[torch.Generator]):
        if batch_size < 1 or batch_size > len(sampler):
            raise ValueError(f"batch_size {batch_size} outside 1..{len(sampler)} speakers")
        self.sampler, self.batch_size, self.shuffle, self.generator = sampler, batch_size, shuffle, generator

    def __len__(self) -> int:
        return len(self.sampler) // self.batch_size

    def __iter__(self) -> Iterator[torch.Tensor]:
        n = len(self.sampler)
        perm: List[int] = torch.randperm(n, generator=self.generator).tolist() if self.shuffle else list(range(n))
        for i in range(0, n - self.batch_size + 1, self.batch_size):
            yield self.sampler.batch(perm[i:i + self.batch_size])


def get_train_test_data_loader(hp):
    """The reference's factory of the same name (s1:80-107): ``(train_loader, test_loader)`` built from ``hp`` alone --
    the ``sv_*.npy`` folders ``hp.m_ge2e.tt_data.train_spects_path`` / ``test_spects_path`` under
    ``hp.general.project_root``, ``training_N`` / ``test_N`` speakers per batch with ``training_M`` / ``test_M`` utterances
    of ``min_train_utter_len`` / ``min_test_utter_len`` frames, the training loader shuffled, both ``drop_last``.  The
    loaders yield (N, M, frames, mels) float32 batches that are already on ``hp.general.device`` (the reference's yield
    float64 host tensors that s4:170 / s5:33 move and s2:28 casts); they are re-iterable, one pass per epoch."""
    root = hp.general.project_root
    dev = hp.general.device
    train_store = SpectrogramStore.from_dir(os.path.join(root, hp.m_ge2e.tt_data.train_spects_path), dev)
    test_store = SpectrogramStore.from_dir(os.path.join(root, hp.m_ge2e.tt_data.test_spects_path), dev)
    train_loader = GE2EBatchSampler.from_hp(train_store, hp, training=True).loader(hp.m_ge2e.training_N, shuffle=True)
    test_loader = GE2EBatchSampler.from_hp(test_store, hp, training=False).loader(hp.m_ge2e.test_N, shuffle=False)
    return train_loader, test_loader
