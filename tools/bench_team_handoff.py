"""Time the team hand-off (ge2e_team.hpp) in isolation: rounds of publish -> signal -> wait -> read all.
Usage (GPU box): python tools/bench_team_handoff.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch  # noqa: E402

from speaker_embedding_ge2e_loss_amd import _lib  # noqa: E402


def main():
    lib = _lib.load()
    dev = "cuda:0"
    grid = torch.cuda.get_device_properties(0).multi_processor_count
    for payload in (64, 512, 4096):
        rounds = 400
        ws = torch.empty(lib.ge2e_selftest_team_bytes(payload) + 256, dtype=torch.uint8, device=dev)
        out = torch.zeros(16, dtype=torch.int32, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        for rep in range(2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = lib.ge2e_selftest_team(ws.data_ptr(), ws.numel(), grid, rounds, payload, out.data_ptr(), st)
            e1.record()
            torch.cuda.synchronize()
            assert rc == 0, rc
        o = out.cpu().numpy()
        us = e0.elapsed_time(e1) * 1e3 / rounds
        print(f"payload {payload * 16 // 1024:3d} KiB/member: {us:6.2f} us per round (two hand-offs + 8 reads), "
              f"teams {o[0]}, per XCD {list(o[2:10])}, bad {o[1]}, abort {o[10]}", flush=True)


if __name__ == "__main__":
    main()
