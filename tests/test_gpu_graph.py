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


def test_workspace_cache_and_capture_on_one_stream():
    """ADVICE (round 3): eager warm-up and capture on the SAME stream, then a larger eager call on it (which replaces the
    cached workspace), then replays.  A graph that had baked in the cached workspace pointer would now write its team
    control block and exchange area into memory the allocator has handed out again; the workspace of a captured launch
    must come from the graph's own pool instead.  Also: two graphs captured one after the other must not share one."""
    from speaker_embedding_ge2e_loss_amd import functional as GF

    dev = torch.device("cuda:0")
    w, b = torch.tensor(10.0, device=dev), torch.tensor(-5.0, device=dev)
    e1 = torch.as_tensor(orc.synth_embeddings((2, 64, 10, 256), "unit", seed=41), device=dev)
    e2 = torch.as_tensor(orc.synth_embeddings((2, 64, 10, 256), "unit", seed=42), device=dev)
    big = torch.as_tensor(orc.synth_embeddings((300, 64, 10, 256), "unit", seed=43), device=dev)
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    outs, graphs = [], []
    with torch.cuda.stream(side):
        for e in (e1, e2):
            for _ in range(2):
                GF.loss_fwd_bwd(e, w, b, impl="team")                      # eager warm-up: fills the (device, stream) cache
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                o = GF.loss_fwd_bwd(e, w, b, impl="team")
            graphs.append(g)
            outs.append(o)
        ref = [GF.loss_fwd_bwd(e, w, b, impl="team") for e in (e1, e2)]
        GF.loss_fwd_bwd(big, w, b, impl="team")                            # larger need on the same stream: cache entry replaced
        junk = [torch.full((1 << 20,), float("nan"), device=dev) for _ in range(8)]   # whatever was freed gets reused
        for _ in range(3):
            for g in graphs:
                g.replay()
        side.synchronize()
    del junk
    for o, r in zip(outs, ref):
        assert torch.equal(o.loss, r.loss) and torch.equal(o.dE, r.dE) and torch.equal(o.dw, r.dw)
    assert len(GF._ws_cache) <= GF._WS_CACHE_MAX


# ---- GE2ELoss(hp, graph=True): the drop-in module's own route to the replayed step ------------------

def _eager_step(eager, E, dev):
    a = torch.as_tensor(E, device=dev).requires_grad_(True)
    eager.zero_grad(set_to_none=True)
    ref = eager(a)
    ref.backward()
    return ref.detach(), a.grad, eager.w.grad, eager.b.grad


@pytest.mark.parametrize("shape", [(64, 10, 256), (2, 16, 256), (4, 5, 256), (24, 6, 128), (256, 4, 128)])
def test_module_graph_route_equals_eager(shape):
    """mod = GE2ELoss(hp, graph=True); mod(e).backward() -- the reference's two lines (s4:196, 200) -- gives the eager step's
    bits: the first call of a shape goes through the eager node, the second captures, later ones replay."""
    from speaker_embedding_ge2e_loss_amd import GE2ELoss, HParams

    dev = torch.device("cuda:0")
    mod, eager = GE2ELoss(HParams(device=dev), graph=True), GE2ELoss(HParams(device=dev))
    for it in range(6):
        E = orc.synth_embeddings(shape, "unit", seed=500 + it)
        e = torch.as_tensor(E, device=dev).requires_grad_(True)
        mod.zero_grad(set_to_none=True)
        loss = mod(e)
        loss.backward()
        ref, ge, gw, gb = _eager_step(eager, E, dev)
        torch.cuda.synchronize()
        assert (len(mod._steps) == 1) == (it >= 1), it
        assert torch.equal(loss.detach(), ref), (it, float(loss), float(ref))
        assert torch.equal(e.grad, ge) and torch.equal(mod.w.grad, gw) and torch.equal(mod.b.grad, gb), it
        assert mod.w.grad.shape == mod.w.shape and e.grad.shape == e.shape
    r = orc.closed_form(E, 10.0, -5.0)
    assert abs(float(loss) - float(r["loss"])) <= 2e-5 * abs(float(r["loss"]))
    assert np.linalg.norm(e.grad.cpu().numpy().astype(np.float64) - r["dE"]) / np.linalg.norm(r["dE"]) <= 2e-5


def test_module_graph_route_accumulates_and_follows_parameter_updates():
    """.grad tensors that are still attached at the next forward are not overwritten by the replay (gradient accumulation over
    two steps = the eager sums), an SGD step on w, b is seen by the next replay (the graph holds their addresses), and a
    gradient that flows THROUGH the loss -- (2 loss).backward(), backward(gradient=...) -- takes the autograd node."""
    from speaker_embedding_ge2e_loss_amd import GE2ELoss, HParams

    dev = torch.device("cuda:0")
    shape = (64, 10, 256)
    mod, eager = GE2ELoss(HParams(device=dev), graph=True), GE2ELoss(HParams(device=dev))
    Es = [orc.synth_embeddings(shape, "unit", seed=600 + i) for i in range(4)]
    for E in Es[:2]:                                       # capture
        mod(torch.as_tensor(E, device=dev).requires_grad_(True)).backward()
    mod.zero_grad(set_to_none=True)
    # accumulation over two steps, no zero_grad in between
    for E in Es[2:]:
        mod(torch.as_tensor(E, device=dev).requires_grad_(True)).backward()
    eager.zero_grad(set_to_none=True)
    for E in Es[2:]:
        eager(torch.as_tensor(E, device=dev).requires_grad_(True)).backward()
    torch.cuda.synchronize()
    assert torch.allclose(mod.w.grad, eager.w.grad, rtol=1e-6, atol=0) and torch.allclose(mod.b.grad, eager.b.grad, rtol=1e-6, atol=1e-9)
    # zero_grad(set_to_none=False) keeps the static buffers attached: still right
    for m in (mod, eager):
        m.zero_grad(set_to_none=False)
        m(torch.as_tensor(Es[0], device=dev).requires_grad_(True)).backward()
    assert torch.equal(mod.w.grad, eager.w.grad) and torch.equal(mod.b.grad, eager.b.grad)
    # a parameter update in place
    with torch.no_grad():
        for m in (mod, eager):
            m.w.add_(-0.5)
            m.b.add_(0.25)
    e1, e2 = (torch.as_tensor(Es[1], device=dev).requires_grad_(True) for _ in range(2))
    mod.zero_grad(set_to_none=True); eager.zero_grad(set_to_none=True)
    l1, l2 = mod(e1), eager(e2)
    (2.0 * l1).backward()                                  # through the loss: the node's backward scales the static gradients
    (2.0 * l2).backward()
    torch.cuda.synchronize()
    assert torch.equal(l1.detach(), l2.detach())
    assert torch.equal(e1.grad, e2.grad) and torch.equal(mod.w.grad, eager.w.grad) and torch.equal(mod.b.grad, eager.b.grad)
    e1.grad = None; mod.zero_grad(set_to_none=True)
    l1 = mod(e1)
    l1.backward(gradient=torch.tensor(2.0, device=dev))
    assert torch.equal(e1.grad, e2.grad)
    # a stale loss says so instead of handing out another step's gradients
    la = mod(e1)
    mod(e1)
    with pytest.raises(RuntimeError, match="earlier forward"):
        (la * 1.0).backward()


def test_module_graph_route_behind_an_encoder_and_across_shapes():
    """Embeddings that come out of a network (not a leaf): loss.backward() hands dE to the engine for the part of the graph
    behind them.  Shapes alternate: each is captured when it comes twice in a row, others go through the eager node."""
    from speaker_embedding_ge2e_loss_amd import GE2ELoss, HParams

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    lin_a = torch.nn.Linear(40, 256).to(dev)
    lin_b = torch.nn.Linear(40, 256).to(dev)
    lin_b.load_state_dict(lin_a.state_dict())
    mod, eager = GE2ELoss(HParams(device=dev), graph=True), GE2ELoss(HParams(device=dev))
    for it, (n, m) in enumerate([(64, 10), (64, 10), (64, 10), (8, 6), (64, 10), (8, 6), (8, 6), (8, 6)]):
        x = torch.randn(n * m, 40, device=dev)
        outs = []
        for lin, loss_mod in ((lin_a, mod), (lin_b, eager)):
            lin.zero_grad(set_to_none=True); loss_mod.zero_grad(set_to_none=True)
            emb = torch.nn.functional.normalize(lin(x), dim=-1).reshape(n, m, 256)
            loss = loss_mod(emb)
            loss.backward()
            outs.append((loss.detach().clone(), lin.weight.grad.clone(), loss_mod.w.grad.clone()))
        torch.cuda.synchronize()
        assert torch.equal(outs[0][0], outs[1][0]), it
        assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2]), it
    assert len(mod._steps) == 2


def test_module_graph_route_latency():
    """What the route is for: the reference's step at one batch per call is host-bound through the eager node; served from
    the static graph it is device-bound (the bench line's latency_module_b1_us is the figure of record; how slow the eager
    step is depends on the process -- 49 us in the middle of this suite, 85-123 us in bench.py's -- so only the order is held)."""
    from speaker_embedding_ge2e_loss_amd import GE2ELoss, HParams

    dev = torch.device("cuda:0")
    E = torch.as_tensor(orc.synth_embeddings((64, 10, 256), "unit", seed=1), device=dev)
    med = {}
    for graph in (False, True):
        mod = GE2ELoss(HParams(device=dev), graph=graph)
        em = E.clone().requires_grad_(True)

        def step():
            em.grad = None
            mod.zero_grad(set_to_none=True)
            mod(em).backward()

        for _ in range(20):
            step()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(101)]
        ev[0].record()
        for i in range(100):
            step()
            ev[i + 1].record()
        torch.cuda.synchronize()
        med[graph] = float(np.median([ev[i].elapsed_time(ev[i + 1]) for i in range(100)])) * 1e3
    print(f"module step: eager {med[False]:.1f} us, graph route {med[True]:.1f} us")
    assert med[True] < med[False] and med[True] < 70.0, med
