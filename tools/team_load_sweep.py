"""How much of the team kernel's time per batch is contention between teams (L2, fabric, HBM) and how much is the team's own
structure?  The same launch capped to 64, 128, 192 and 256 workgroups (1..4 teams per XCD), the same number of batches PER
TEAM, cycles per batch and team from HIP events.  Usage: python tools/team_load_sweep.py [per_team]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from speaker_embedding_ge2e_loss_amd import _lib, functional as GF  # noqa: E402


def main():
    per_team = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    N, M, D = 64, 10, 256
    lib = _lib.load()
    dev = torch.device("cuda:0")
    w, b = torch.tensor(10.0, device=dev), torch.tensor(-5.0, device=dev)
    g = torch.Generator(device=dev).manual_seed(3)
    for cap in (64, 128, 192, 256):
        teams = cap // 8
        B = teams * per_team
        e = torch.randn(B, N, M, D, device=dev, generator=g)
        e = e / e.norm(dim=-1, keepdim=True)
        o = GF.LossOutputs(loss=torch.empty(B, device=dev), per=None, dE=torch.empty(B, N, M, D, device=dev),
                           dw=torch.empty(B, device=dev), db=torch.empty(B, device=dev))
        ws = GF.alloc_workspace(GF.workspace_bytes(B, N, M, D, "softmax", "team"), dev)
        st = torch.cuda.current_stream().cuda_stream

        def call():
            rc = lib.ge2e_selftest_team_grid(e.data_ptr(), B, N, M, D, w.data_ptr(), b.data_ptr(), 1e-8, 1e-6, 0,
                                             o.loss.data_ptr(), None, o.dE.data_ptr(), o.dw.data_ptr(), o.db.data_ptr(),
                                             ws.data_ptr(), ws.numel(), st, cap)
            assert rc == 0

        for _ in range(3):
            call()
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(11)]
        ev[0].record()
        for i in range(10):
            call()
            ev[i + 1].record()
        torch.cuda.synchronize()
        ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(10))[5]
        print(f"cap {cap:3d} workgroups = {teams:2d} teams, B = {B:5d} ({per_team} per team): {ms * 1e3:8.1f} us per launch, "
              f"{ms * 1e3 / per_team:6.2f} us per batch and team, {B / ms * 1e3 / 1e6:5.2f} M batches/s", flush=True)


if __name__ == "__main__":
    main()
