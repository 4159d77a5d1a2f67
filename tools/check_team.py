"""Quick parity probe of one implementation against the oracle (development aid).
Usage (GPU box): python tools/check_team.py [impl] [B N M D] [variant]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import ge2e_oracle as orc  # noqa: E402
from speaker_embedding_ge2e_loss_amd import functional as GF  # noqa: E402


def main():
    impl = sys.argv[1] if len(sys.argv) > 1 else "team"
    shape = tuple(int(x) for x in sys.argv[2:6]) if len(sys.argv) > 5 else (3, 64, 10, 256)
    variant = sys.argv[6] if len(sys.argv) > 6 else "softmax"
    E = orc.synth_embeddings(shape, "unit", seed=5)
    dev = torch.device("cuda:0")
    e = torch.as_tensor(E, device=dev)
    w, b = torch.tensor(10.0, device=dev), torch.tensor(-5.0, device=dev)
    nan = lambda *s: torch.full(s, float("nan"), device=dev)  # noqa: E731
    out = GF.LossOutputs(loss=nan(shape[0]), per=nan(*shape[:3]), dE=nan(*shape), dw=nan(shape[0]), db=nan(shape[0]))
    o = GF.loss_fwd_bwd(e, w, b, variant=variant, impl=impl, out=out)
    torch.cuda.synchronize()
    for bi in range(shape[0]):
        ref = orc.closed_form(E[bi], 10.0, -5.0, variant=variant)
        dE = o.dE[bi].cpu().numpy()
        print(f"batch {bi}: loss {float(o.loss[bi]):.6f} ref {ref['loss']:.6f} | dw {float(o.dw[bi]):.6f} ref {ref['dw']:.6f} | "
              f"db {float(o.db[bi]):.3e} ref {ref['db']:.3e} | per max err {np.abs(o.per[bi].cpu().numpy() - ref['per']).max():.2e} | "
              f"dE rel fro {np.linalg.norm(dE - ref['dE']) / np.linalg.norm(ref['dE']):.2e} nan {np.isnan(dE).sum()}", flush=True)
        if bi == 0:
            err = np.abs(dE - ref["dE"]).max(axis=2)
            print("  per-row max err of batch 0, speakers 0..2:\n", np.array2string(err[:3], precision=2))


if __name__ == "__main__":
    main()
