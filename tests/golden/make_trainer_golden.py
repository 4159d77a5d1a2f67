"""Capture SGD loss trajectories from the reference's own encoder + loss (build container only).

    python tests/golden/make_trainer_golden.py

Imports ``embedding_model_GE2E/s2_model_GE2E_loss_speach_embed.py`` (the encoder) and
``s3_loss_function_GE2E.py`` (the loss) from /root/reference and drives them with the statement
sequence of the reference's training loop -- ``s4_train_embed_model.py:167-205`` (reshape, random
permutation, encoder, un-permute, loss, zero_grad, backward, the two clips, step), ``:261-264`` (LR
halving of group 0 only) and ``:61-110`` (batched test loss in eval mode).  ``TrainEmbedModel``
itself cannot be constructed here (its ``__init__`` opens the spectrogram folders, s4:45), so the
loop is driven from this script; the modules that do the arithmetic are the reference's.

Written to ``tests/golden/callers/trainer_*.npz``: the initial weights, every input batch, every
permutation drawn, the loss after each step, the final weights and the test loss.  Data only.
"""
import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, REF)

from embedding_model_GE2E.s2_model_GE2E_loss_speach_embed import ModelGE2ELossSpeachEmbed  # noqa: E402
from embedding_model_GE2E.s3_loss_function_GE2E import GE2ELoss  # noqa: E402
from utils.dict_to_dot import GetDictWithDotNotation  # noqa: E402


def capture(name, n_mels, hidden, layers, emb, N, M, T, steps=5, lr=0.05, halve_after=3, seed=11,
            n_test=2):
    torch.set_num_threads(1)
    hp = GetDictWithDotNotation({
        "general": {"small_err": 1e-6, "device": torch.device("cpu")},
        "audio": {"mel_n_channels": n_mels},
        "m_ge2e": {"model_hidden_size": hidden, "model_embedding_size": emb, "model_num_layers": layers},
    })
    torch.manual_seed(seed)
    model = ModelGE2ELossSpeachEmbed(hp)
    loss_mod = GE2ELoss(hp)
    opt = torch.optim.SGD([{"params": model.parameters()}, {"params": loss_mod.parameters()}], lr=lr)  # s4:35-42
    out = {"cfg": np.array([n_mels, hidden, layers, emb, N, M, T, steps, halve_after, seed, n_test], dtype=np.int64),
           "lr": np.float64(lr)}
    for k, v in model.state_dict().items():
        out["init." + k] = v.numpy().copy()
    g = torch.Generator().manual_seed(seed + 1)
    mels = torch.randn(steps, N, M, T, n_mels, generator=g)
    test_mels = torch.randn(n_test, N, M, T, n_mels, generator=g)
    out["mels"] = mels.numpy()
    out["test_mels"] = test_mels.numpy()

    total = N * M
    random.seed(seed)
    cur_lr = lr
    losses, perms = [], []
    model.train()
    for s in range(steps):
        mel = torch.reshape(mels[s], (total, T, n_mels))                  # s4:170-171
        perm = random.sample(range(0, total), total)                      # s4:174
        unperm = list(perm)
        for i, j in enumerate(perm):                                      # s4:179-180
            unperm[j] = i
        e = model(mel[perm])[unperm]                                      # s4:182-186
        e = torch.reshape(e, (N, M, e.size(1)))                           # s4:189
        loss = loss_mod(e)                                                # s4:193
        opt.zero_grad()                                                   # s4:196-200
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 3.0)
        torch.nn.utils.clip_grad_norm_(loss_mod.parameters(), 1.0)
        opt.step()
        losses.append(float(loss.detach()))
        perms.append(perm)
        if s + 1 == halve_after:                                          # s4:261-264
            cur_lr = cur_lr / 2
            opt.param_groups[0]["lr"] = cur_lr

    model.eval()                                                          # s4:69
    test_losses, test_perms = [], []
    for t in range(n_test):
        mel = torch.reshape(test_mels[t], (total, T, n_mels))
        perm = random.sample(range(0, total), total)
        unperm = list(perm)
        for i, j in enumerate(perm):
            unperm[j] = i
        e = model(mel[perm])[unperm]
        e = torch.reshape(e, (N, M, e.shape[1]))
        test_losses.append(loss_mod(e).to("cpu").detach().numpy())       # s4:103-104
        test_perms.append(perm)
    model.train()

    out["losses"] = np.array(losses, dtype=np.float64)
    out["perms"] = np.array(perms, dtype=np.int64)
    out["test_perms"] = np.array(test_perms, dtype=np.int64)
    out["test_loss_mean"] = np.float64(np.mean(test_losses))              # s4:109
    out["final_lrs"] = np.array([g_["lr"] for g_ in opt.param_groups], dtype=np.float64)
    for k, v in model.state_dict().items():
        out["final." + k] = v.numpy().copy()
    out["final_w"] = loss_mod.w.detach().numpy().copy()
    out["final_b"] = loss_mod.b.detach().numpy().copy()
    path = os.path.join(HERE, "callers", name + ".npz")
    np.savez(path, **out)
    print(f"{name}: losses {['%.5f' % x for x in losses]} test {float(out['test_loss_mean']):.5f} "
          f"w {float(out['final_w']):.6f} b {float(out['final_b']):.6f} ({os.path.getsize(path) / 1024:.0f} KiB)")


if __name__ == "__main__":
    # tiny: the reference's own shape family (few speakers), runs on the generic kernel
    capture("trainer_tiny", n_mels=8, hidden=16, layers=2, emb=16, N=4, M=5, T=12)
    # 16 speakers, D=64: the shape class the team kernel takes under impl="auto"
    capture("trainer_n16_d64", n_mels=8, hidden=32, layers=3, emb=64, N=16, M=6, T=10, seed=23)
