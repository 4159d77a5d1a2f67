"""Engine clock and power while the team kernel runs, at 1/4 and at full occupancy of the chip (rocm-smi sampled from a
second process while this one keeps the queue full).  Usage: python tools/clock_under_load.py"""
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from speaker_embedding_ge2e_loss_amd import _lib, functional as GF  # noqa: E402


def smi():
    out = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showclocks", "--showpower", "-d", "0"], capture_output=True, text=True).stdout
    keep = [ln.strip() for ln in out.splitlines() if "sclk" in ln or "Power" in ln or "mclk" in ln or "fclk" in ln]
    return " | ".join(keep)


def main():
    N, M, D = 64, 10, 256
    lib = _lib.load()
    dev = torch.device("cuda:0")
    w, b = torch.tensor(10.0, device=dev), torch.tensor(-5.0, device=dev)
    print("idle:", smi(), flush=True)
    for cap in (64, 256):
        B = cap // 8 * 128
        e = torch.randn(B, N, M, D, device=dev)
        e = e / e.norm(dim=-1, keepdim=True)
        o = GF.LossOutputs(loss=torch.empty(B, device=dev), per=None, dE=torch.empty(B, N, M, D, device=dev),
                           dw=torch.empty(B, device=dev), db=torch.empty(B, device=dev))
        ws = GF.alloc_workspace(GF.workspace_bytes(B, N, M, D, "softmax", "team"), dev)
        st = torch.cuda.current_stream().cuda_stream
        t_end = time.time() + 6.0
        k = 0
        while time.time() < t_end:
            for _ in range(50):
                lib.ge2e_selftest_team_grid(e.data_ptr(), B, N, M, D, w.data_ptr(), b.data_ptr(), 1e-8, 1e-6, 0,
                                            o.loss.data_ptr(), None, o.dE.data_ptr(), o.dw.data_ptr(), o.db.data_ptr(),
                                            ws.data_ptr(), ws.numel(), st, cap)
            k += 1
            if k % 8 == 0:
                print(f"cap {cap}:", smi(), flush=True)
            torch.cuda.synchronize()


if __name__ == "__main__":
    main()
