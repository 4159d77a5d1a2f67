"""DPTrainer against loss trajectories captured from the reference's own encoder + loss
(tests/golden/make_trainer_golden.py: reference s2 + s3 driven by s4's statement sequence).

CPU leg: the trainer's host logic (perm/unperm stream, clips, two param groups, LR halving, eval
loss, encoder-only checkpoint) with the ORACLE's expand form as the loss -- the product loss has no
CPU path.  GPU leg: the same trajectories with the HIP ``GE2ELoss`` on the device.
"""
import os

import numpy as np
import pytest
import torch

from speaker_embedding_ge2e_loss_amd.encoder import SpeakerEncoder
from speaker_embedding_ge2e_loss_amd.trainer import DPTrainer

GOLD = os.path.join(os.path.dirname(__file__), "golden", "callers")
FIXTURES = ["trainer_tiny", "trainer_n16_d64"]


def load(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    n_mels, hidden, layers, emb, N, M, T, steps, halve_after, seed, n_test = [int(v) for v in z["cfg"]]
    return z, dict(n_mels=n_mels, hidden=hidden, layers=layers, emb=emb, N=N, M=M, T=T, steps=steps,
                   halve_after=halve_after, seed=seed, n_test=n_test, lr=float(z["lr"]))


def encoder_from(z, c, device, normalize=True):
    enc = SpeakerEncoder(c["n_mels"], c["hidden"], c["layers"], c["emb"], normalize=normalize)
    state = {k[len("init."):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("init.")}
    missing = enc.load_state_dict(state, strict=True)  # the reference's parameter names load as they are
    assert not missing.missing_keys and not missing.unexpected_keys
    return enc.to(device)


def run_trajectory(z, c, trainer, device):
    losses = []
    for s in range(c["steps"]):
        losses.append(trainer.step(torch.from_numpy(z["mels"][s]).to(device)))
        if s + 1 == c["halve_after"]:
            trainer.halve_lr()
    test = trainer.eval_loss([torch.from_numpy(m).to(device) for m in z["test_mels"]])
    return [float(x) for x in losses], test


def check(z, c, trainer, losses, test, rtol):
    assert np.allclose(losses, z["losses"], rtol=rtol, atol=0), (losses, z["losses"])
    assert abs(test - float(z["test_loss_mean"])) <= rtol * abs(float(z["test_loss_mean"]))
    assert [g["lr"] for g in trainer.optimizer.param_groups] == list(z["final_lrs"])
    final = trainer.model.state_dict()
    for k in z.files:
        if k.startswith("final."):
            got = final[k[len("final."):]].detach().cpu().numpy()
            assert np.allclose(got, z[k], rtol=20 * rtol, atol=2e-6), (k, np.abs(got - z[k]).max())
    assert abs(float(trainer.ge2e_loss.w) - float(z["final_w"])) < 5e-5
    assert abs(float(trainer.ge2e_loss.b) - float(z["final_b"])) < 5e-5


@pytest.mark.parametrize("name", FIXTURES)
def test_trainer_reproduces_reference_trajectory_cpu(name, tmp_path):
    from oracle import ge2e_oracle as orc

    class OracleLoss(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.tensor(10.0))
            self.b = torch.nn.Parameter(torch.tensor(-5.0))

        def forward(self, emb):
            return orc.expand_form_loss(emb, self.w, self.b)[0]

    torch.set_num_threads(1)
    z, c = load(name)
    tr = DPTrainer(encoder_from(z, c, "cpu"), OracleLoss(), lr=c["lr"], seed=c["seed"])
    # the permutation stream is the reference's: random.seed(s) + random.sample per batch (s4:174)
    import random
    probe = random.Random(c["seed"])
    assert probe.sample(range(c["N"] * c["M"]), c["N"] * c["M"]) == list(z["perms"][0])
    losses, test = run_trajectory(z, c, tr, "cpu")
    check(z, c, tr, losses, test, rtol=2e-6)

    # s4:130: the checkpoint is the encoder alone and loads into a fresh encoder
    path = str(tmp_path / "ckpt.pth")
    tr.save_checkpoint(path)
    state = torch.load(path)
    assert set(state) == {k[len("final."):] for k in z.files if k.startswith("final.")}
    assert not any(k in state for k in ("w", "b")) and all(v.device.type == "cpu" for v in state.values())
    fresh = SpeakerEncoder(c["n_mels"], c["hidden"], c["layers"], c["emb"])
    fresh.load_state_dict(state)
    for k, v in fresh.state_dict().items():
        assert np.array_equal(v.numpy(), tr.model.state_dict()[k].numpy())
    assert tr.model.training  # eval_loss put the encoder back in train mode (s4:107)


def test_foreign_zero_grad_does_not_detach_the_bucket():
    """optimizer.zero_grad() (set_to_none by default) drops the .grad views; the next step must
    still reduce and apply the bucket."""
    z, c = load("trainer_tiny")
    from tests.test_trainer_gloo import OracleLoss
    tr = DPTrainer(encoder_from(z, c, "cpu"), OracleLoss(), lr=c["lr"], seed=c["seed"])
    tr.optimizer.zero_grad()
    assert all(p.grad is None for p in tr._params)
    before = [p.detach().clone() for p in tr._params]
    tr.step(torch.from_numpy(z["mels"][0]))
    base = tr.flat_grad.untyped_storage().data_ptr()
    assert all(p.grad.untyped_storage().data_ptr() == base for p in tr._params)
    assert float(tr.flat_grad.abs().sum()) > 0
    assert any(not torch.equal(a, p.detach()) for a, p in zip(before, tr._params))


def test_fit_loop_halves_checkpoints_and_evaluates(tmp_path):
    z, c = load("trainer_tiny")
    from tests.test_trainer_gloo import OracleLoss
    tr = DPTrainer(encoder_from(z, c, "cpu"), OracleLoss(), lr=c["lr"], seed=c["seed"])
    train = [torch.from_numpy(z["mels"][s]) for s in range(2)]
    test = [torch.from_numpy(m) for m in z["test_mels"]]
    _, tl, vl = tr.fit(train, epochs=4, test_batches=test, lr_reduce=2, epoch_print=2,
                       checkpoint_dir=str(tmp_path), checkpoint_interval=3)
    assert len(tl) == 4 and len(vl) == 2
    assert [g["lr"] for g in tr.optimizer.param_groups] == [c["lr"] / 4, c["lr"]]
    names = sorted(os.listdir(tmp_path))
    assert len(names) == 2 and names[0].startswith("ckpt_epoch_3_L_") and names[1].startswith("final_epoch_4_L_")


def test_mixed_dtype_parameters_are_refused():
    enc = SpeakerEncoder(4, 4, 1, 4)
    enc.projection.double()
    from tests.test_trainer_gloo import OracleLoss
    with pytest.raises(ValueError, match="one device and dtype"):
        DPTrainer(enc, OracleLoss())


@pytest.mark.gpu
@pytest.mark.parametrize("name", FIXTURES)
def test_trainer_with_the_graph_route_loss_on_gpu(name):
    """GE2ELoss(hp, graph=True) inside the trainer: the embeddings come out of the encoder (not a leaf), the parameters'
    gradients are views of the flat bucket (the shortcut adds into them), the test loss runs under no_grad (eager node)."""
    from speaker_embedding_ge2e_loss_amd import GE2ELoss, HParams
    dev = torch.device("cuda:0")
    z, c = load(name)
    loss = GE2ELoss(HParams(device=dev), graph=True)
    tr = DPTrainer(encoder_from(z, c, dev), loss, lr=c["lr"], seed=c["seed"])
    losses, test = run_trajectory(z, c, tr, dev)
    check(z, c, tr, losses, test, rtol=1e-4)
    assert len(loss._steps) == 1                       # captured on the second step, replayed from then on


@pytest.mark.gpu
@pytest.mark.parametrize("name", FIXTURES)
@pytest.mark.parametrize("fused_tail", [False, True])
def test_trainer_with_hip_loss_on_gpu(name, fused_tail):
    """The same trajectories with the product: encoder on the device (MIOpen/rocBLAS LSTM), the HIP
    GE2ELoss as the loss module.  Tolerance 1e-4 relative on every loss of the trajectory
    (north_star's bound for fp32 results), looser on the weights after 5 steps."""
    from speaker_embedding_ge2e_loss_amd import GE2ELoss, HParams
    dev = torch.device("cuda:0")
    z, c = load(name)
    enc = encoder_from(z, c, dev, normalize=not fused_tail)
    loss = GE2ELoss(HParams(device=dev))
    tr = DPTrainer(enc, loss, lr=c["lr"], seed=c["seed"], fused_tail=fused_tail)
    losses, test = run_trajectory(z, c, tr, dev)
    check(z, c, tr, losses, test, rtol=1e-4)


def test_fit_refuses_a_one_shot_generator():
    """fit() iterates train_batches once per epoch: a generator is empty from the second epoch on, which used to become
    NaN epoch means and a checkpoint named after them.  Now it raises; a re-iterable (a list, GE2EBatchSampler.loader)
    trains every epoch."""
    z, c = load("trainer_tiny")
    from tests.test_trainer_gloo import OracleLoss
    tr = DPTrainer(encoder_from(z, c, "cpu"), OracleLoss(), lr=c["lr"], seed=c["seed"])
    one_shot = (torch.from_numpy(z["mels"][s]) for s in range(2))
    with pytest.raises(ValueError, match="yielded no batch"):
        tr.fit(one_shot, epochs=2)


def test_fit_saves_the_best_test_loss_on_request(tmp_path):
    z, c = load("trainer_tiny")
    from tests.test_trainer_gloo import OracleLoss
    tr = DPTrainer(encoder_from(z, c, "cpu"), OracleLoss(), lr=c["lr"], seed=c["seed"])
    train = [torch.from_numpy(z["mels"][s]) for s in range(2)]
    test = [torch.from_numpy(m) for m in z["test_mels"]]
    _, _, vl = tr.fit(train, epochs=3, test_batches=test, epoch_print=1, checkpoint_dir=str(tmp_path),
                      checkpoint_interval=100, save_best_weights=True)
    best = [n for n in os.listdir(tmp_path) if n.startswith("m_best_")]
    # one file per epoch whose test loss was the best so far (s4:243-254); the first always is
    running, expect = None, 0
    for v in vl:
        if running is None or v <= running:
            running, expect = v, expect + 1
    assert len(best) == expect >= 1
