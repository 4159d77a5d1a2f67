"""Development aid: per-batch / per-row parity of one implementation on a test-style problem, run twice.
Usage (GPU box): python tools/probe_impl.py impl B N M D [variant] [kind] [w] [b]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import ge2e_oracle as orc  # noqa: E402
from speaker_embedding_ge2e_loss_amd import functional as GF  # noqa: E402


def main():
    impl = sys.argv[1]
    shape = tuple(int(x) for x in sys.argv[2:6])
    variant = sys.argv[6] if len(sys.argv) > 6 else "softmax"
    kind = sys.argv[7] if len(sys.argv) > 7 else "raw"
    w0 = float(sys.argv[8]) if len(sys.argv) > 8 else 6.0
    b0 = float(sys.argv[9]) if len(sys.argv) > 9 else -1.5
    E = orc.synth_embeddings(shape, kind, seed=sum(shape))
    ref = orc.closed_form(E, w0, b0, variant=variant)
    dev = torch.device("cuda:0")
    e = torch.as_tensor(E, device=dev)
    w, b = torch.tensor(w0, device=dev), torch.tensor(b0, device=dev)
    nan = lambda *s: torch.full(s, float("nan"), device=dev)  # noqa: E731
    for rep in range(2):
        out = GF.LossOutputs(loss=nan(shape[0]), per=nan(*shape[:3]), dE=nan(*shape), dw=nan(shape[0]), db=nan(shape[0]))
        o = GF.loss_fwd_bwd(e, w, b, variant=variant, impl=impl, out=out)
        torch.cuda.synchronize()
        dE = o.dE.cpu().numpy()
        bad = []
        for bi in range(shape[0]):
            rf = np.linalg.norm(dE[bi] - ref["dE"][bi]) / np.linalg.norm(ref["dE"][bi])
            lerr = abs(float(o.loss[bi]) - ref["loss"][bi]) / abs(ref["loss"][bi])
            if not (rf < 2e-5 and lerr < 2e-5):
                bad.append((bi, rf, lerr))
        print(f"rep {rep}: {len(bad)} bad batches of {shape[0]}: {[(b_, f'{r:.1e}', f'{l:.1e}') for b_, r, l in bad[:12]]}", flush=True)
        for bi, _, _ in bad[:2]:
            err = np.abs(dE[bi] - ref["dE"][bi]).max(axis=2) / np.abs(ref["dE"][bi]).max()
            print(f"  batch {bi}: per-row max err / max|dE| (rows = speakers):\n", np.array2string(err, precision=1, max_line_width=200))
            cerr = np.abs(dE[bi] - ref["dE"][bi]).max(axis=(0, 1))
            print("  per-column max err, 16-column tiles:", np.array2string(cerr.reshape(-1, 16).max(axis=1), precision=1, max_line_width=200))


if __name__ == "__main__":
    main()
