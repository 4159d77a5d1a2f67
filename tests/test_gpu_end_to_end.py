"""The callers either side of the loss, chained as the reference chains them (s1 loader -> s4 training step -> s4 test
loss -> s5 EER), every hot piece on the GPU: resident spectrogram store + sampler kernel, encoder (torch LSTM), fused
encoder tail, HIP GE2E loss + backward, flat-bucket trainer, EER sweep kernel."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_train_eval_pipeline_on_synthetic_speakers(capsys):
    from speaker_embedding_ge2e_loss_amd import GE2ELoss, HParams
    from speaker_embedding_ge2e_loss_amd.data import GE2EBatchSampler, SpectrogramStore
    from speaker_embedding_ge2e_loss_amd.encoder import SpeakerEncoder
    from speaker_embedding_ge2e_loss_amd.evaluation import calculate_ERR
    from speaker_embedding_ge2e_loss_amd.trainer import DPTrainer

    dev = torch.device("cuda:0")
    rng = np.random.default_rng(3)
    T, F, S = 24, 8, 12
    # every speaker = its own spectral signature + noise: separable, so a few steps must lower the loss
    sig = rng.standard_normal((S, 1, 1, F)) * 2.0
    arrays = [(sig[j] + 0.3 * rng.standard_normal((6, T, F))).astype(np.float64) for j in range(S)]
    store = SpectrogramStore(arrays, [f"sv_{j}.npy" for j in range(S)], dev)
    np.random.seed(0)
    train = GE2EBatchSampler(store, utter_num=5, min_utter_len=16, training=False)
    gen = torch.Generator().manual_seed(0)

    torch.manual_seed(0)
    enc = SpeakerEncoder(F, 24, 2, 16, normalize=False).to(dev)
    hp = HParams(device=dev)
    tr = DPTrainer(enc, GE2ELoss(hp), lr=0.05, seed=1, fused_tail=True)
    first = last = None
    for epoch in range(6):
        for mel in train.loader(batch_size=4, shuffle=True, generator=gen):   # (4, 5, 16, 8) float32 on the device
            loss = float(tr.step(mel))
            assert np.isfinite(loss)
            first = loss if first is None else first
            last = loss
    assert last < first, (first, last)

    test_batches = list(train.loader(batch_size=4, shuffle=False))
    ev = tr.eval_loss(test_batches)
    assert np.isfinite(ev) and tr.model.training

    # s5: EER on a test batch through the trained encoder (normalising encoder for evaluation: same weights)
    enc_eval = SpeakerEncoder(F, 24, 2, 16, normalize=True).to(dev)
    enc_eval.load_state_dict(enc.state_dict())
    res = calculate_ERR(enc_eval.eval(), hp, N=4, M=5, test_loader=[b.reshape(1, 20, 16, F) for b in test_batches[:1]])
    assert "EER :" in capsys.readouterr().out and set(res[0]) == {"EER", "thres", "FAR", "FRR"}


def _hp_for(tmp_path, dev, F=8):
    """An ``hp`` with every field the reference's scripts read on this path (strings/constants.py:29-110), as a plain
    attribute tree (the reference uses a dict with dot access)."""
    from types import SimpleNamespace as NS
    return NS(
        general=NS(small_err=1e-6, device=dev, project_root=str(tmp_path)),
        audio=NS(mel_n_channels=F),
        m_ge2e=NS(tt_data=NS(train_spects_path="spects/train", test_spects_path="spects/test",
                             min_train_utter_len=16, min_test_utter_len=12),
                  model_hidden_size=24, model_embedding_size=64, model_num_layers=2, lr=0.05, training_epochs=4,
                  checkpoint_dir="chk", save_best_weights=True, min_test_loss=1e9, restore_existing_model=False,
                  checkpoint_interval=2, training_N=4, training_M=5, test_N=4, test_M=5))


def test_reference_shaped_entry_points_built_from_hp(tmp_path, capsys):
    """`TrainEmbedModel(hp).train_model(lr_reduce, epoch_print, dot_print)` (s4:19-59, :137-276) and
    `calculate_ERR(model, hp, N, M)` (s5:16-21) with the reference's signatures: loaders, encoder, loss, optimizer and
    checkpoint folder all come from `hp`, the spectrogram folders are the reference's `sv_*.npy` files."""
    import os
    from speaker_embedding_ge2e_loss_amd.evaluation import calculate_ERR
    from speaker_embedding_ge2e_loss_amd.trainer import TrainEmbedModel

    dev = torch.device("cuda:0")
    rng = np.random.default_rng(5)
    T, F = 24, 8
    for split, S in (("train", 12), ("test", 8)):
        d = tmp_path / "spects" / split
        d.mkdir(parents=True)
        sig = rng.standard_normal((S, 1, 1, F)) * 2.0
        for j in range(S):
            np.save(d / f"sv_{j}.npy", (sig[j] + 0.3 * rng.standard_normal((6, T, F))).astype(np.float64))
    hp = _hp_for(tmp_path, dev, F)
    torch.manual_seed(0)
    np.random.seed(0)
    obj = TrainEmbedModel(hp)
    assert obj.total_utterances == 20 and len(obj.optimizer.param_groups) == 2
    assert [n for n, _ in obj.ge2e_loss.named_parameters()] == ["w", "b"]
    model, train_losses, test_losses = obj.train_model(lr_reduce=3, epoch_print=2, dot_print=1)
    assert model is obj.model and len(train_losses) == 4 and len(test_losses) == 2
    assert all(np.isfinite(train_losses)) and train_losses[-1] < train_losses[0]
    assert obj.lr == 0.025 and obj.optimizer.param_groups[0]["lr"] == 0.025 and obj.optimizer.param_groups[1]["lr"] == 0.05
    names = sorted(os.listdir(tmp_path / "chk"))
    assert any(n.startswith("ckpt_epoch_2_") for n in names) and any(n.startswith("final_epoch_4_") for n in names)
    assert any(n.startswith("m_best_") for n in names)
    # s5: the reference's call, no loader argument -- built from hp's test folder with N, M written into hp first
    res = calculate_ERR(model.eval(), hp, N=2, M=4)
    assert hp.m_ge2e.test_N == 2 and hp.m_ge2e.test_M == 4
    assert len(res) == 4 and "EER :" in capsys.readouterr().out        # 8 test speakers, 2 per batch, drop_last
