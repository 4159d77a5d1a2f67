"""The d-vector encoder the loss is trained with (the caller on the input side of the hot path).

Mirrors ``embedding_model_GE2E/s2_model_GE2E_loss_speach_embed.py:7-35``: a stacked LSTM over
(batch, frames, n_mels), the LAST frame's hidden state through one Linear, L2-normalised.  Attribute
names (``LSTM_stack``, ``projection``) and the initialisation (xavier-normal weights, zero biases
for the LSTM, s2:18-22; the Linear keeps torch's default) follow the reference so a reference
``state_dict`` loads here and the other way round (s4:130 saves exactly this module's).

The recurrent and projection GEMMs are library work (MIOpen / hipBLASLt through torch) and are NOT
part of the hand-written path; what is ours is the tail: with ``normalize=False`` the module hands
back the raw projection and the trainer runs the fused L2-normalise + un-permute gather
(``functional.normalize_unperm``) in one HIP kernel instead of three torch ops.
"""
from __future__ import annotations

import torch
import torch.nn as nn


class SpeakerEncoder(nn.Module):
    def __init__(self, n_mels: int = 80, hidden: int = 256, layers: int = 3, embedding: int = 256,
                 normalize: bool = True):
        super().__init__()
        self.LSTM_stack = nn.LSTM(n_mels, hidden, num_layers=layers, batch_first=True)  # s2:13-16
        for name, param in self.LSTM_stack.named_parameters():  # s2:18-22
            if "bias" in name:
                nn.init.constant_(param, 0.0)
            elif "weight" in name:
                nn.init.xavier_normal_(param)
        self.projection = nn.Linear(hidden, embedding)  # s2:25
        self.normalize = normalize
        self.embedding_size = embedding   # what the trainer needs to know before the forward: the loss kernel's D

    @classmethod
    def from_hp(cls, hp, normalize: bool = True):
        """Same constructor argument as the reference's class (s2:9)."""
        return cls(hp.audio.mel_n_channels, hp.m_ge2e.model_hidden_size, hp.m_ge2e.model_num_layers,
                   hp.m_ge2e.model_embedding_size, normalize=normalize).to(hp.general.device)

    def raw(self, x):
        x, _ = self.LSTM_stack(x.float())       # s2:28
        return self.projection(x[:, x.size(1) - 1].float())  # s2:30-31

    def forward(self, x):
        y = self.raw(x)
        if not self.normalize:
            return y
        return y / torch.norm(y, dim=1).unsqueeze(1)  # s2:34


def get_pre_trained_embedding_model(hp, use_path_as_absolute: bool = False) -> SpeakerEncoder:
    """The reference's loader of the same name (s2:38-67): a fresh encoder built from ``hp``, the encoder-only checkpoint
    ``hp.m_ge2e.best_model_path`` (joined with ``hp.general.project_root`` unless ``use_path_as_absolute``) loaded into
    it, eval mode, on ``hp.general.device``.  The file format is the reference's (s4:130: the module's ``state_dict``),
    which is also what ``DPTrainer.save_checkpoint`` writes."""
    import os
    model = SpeakerEncoder.from_hp(hp)
    path = hp.m_ge2e.best_model_path if use_path_as_absolute else os.path.join(hp.general.project_root, hp.m_ge2e.best_model_path)
    model.load_state_dict(torch.load(path, map_location=hp.general.device))
    return model.eval().to(hp.general.device)
