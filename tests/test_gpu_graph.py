"""The module step under a HIP graph (torch.cuda.CUDAGraph): every launch of the library is capture-safe, a replayed step
gives the eager step's bits -- also from the second replay on, where a memset node for the team kernel's control block
used to come back with another node's pattern (the block is zeroed by a kernel now)."""
import numpy as np
import pytest
import torch

from oracle import ge2e_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("direct", [False, True])
@pytest.mark.parametrize("shape", [(64, 10, 256), (2, 16, 256), (4, 5, 256), (24, 6, 128), (256, 4, 128)])
def test_graphed_loss_step_equals_eager(shape, direct):
    from speaker_embedding_ge2e_loss_amd import GE2ELoss, HParams
    from speaker_embedding_ge2e_loss_amd.graphed import GraphedLossStep

    dev = torch.device("cuda:0")
    mod = GE2ELoss(HParams(device=dev))
    step = GraphedLossStep(mod, shape, direct=direct)
    eager = GE2ELoss(HParams(device=dev))
    for it in range(6):                                     # well past the second replay
        E = orc.synth_embeddings(shape, "unit", seed=300 + it)
        e = torch.as_tensor(E, device=dev)
        loss = step(e)
        a = e.clone().requires_grad_(True)
        eager.zero_grad(set_to_none=True)
        ref = eager(a)
        ref.backward()
        torch.cuda.synchronize()
        assert torch.equal(loss, ref.detach()), (it, float(loss), float(ref))
        assert torch.equal(step.input_grad, a.grad), it
        assert torch.equal(mod.w.grad, eager.w.grad) and torch.equal(mod.b.grad, eager.b.grad)
    # and against the oracle, once
    r = orc.closed_form(E, 10.0, -5.0)
    assert abs(float(loss) - float(r["loss"])) <= 2e-5 * abs(float(r["loss"]))
    num = np.linalg.norm(step.input_grad.cpu().numpy().astype(np.float64) - r["dE"])
    assert num / np.linalg.norm(r["dE"]) <= 2e-5


def test_back_to_back_replays_keep_the_teams():
    """200 replays without a host sync in between: the team kernel must keep forming its teams (no fall-back), i.e. the
    replays stay at the team kernel's speed and its results."""
    from speaker_embedding_ge2e_loss_amd import GE2ELoss, HParams
    from speaker_embedding_ge2e_loss_amd.graphed import GraphedLossStep

    dev = torch.device("cuda:0")
    mod = GE2ELoss(HParams(device=dev), impl="team")
    step = GraphedLossStep(mod, (64, 10, 256))
    first = step().clone()
    g0 = step.input_grad.clone()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(201)]
    ev[0].record()
    for i in range(200):
        step()
        ev[i + 1].record()
    torch.cuda.synchronize()
    assert torch.equal(step.loss, first) and torch.equal(step.input_grad, g0)   # bitwise deterministic
    med = float(np.median([ev[i].elapsed_time(ev[i + 1]) for i in range(200)])) * 1e3
    assert med < 90.0, f"{med:.1f} us per replayed step: the one-workgroup-per-batch fall-back alone takes 120 us"
