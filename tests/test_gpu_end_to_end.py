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
