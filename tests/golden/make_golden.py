"""Generate golden vectors by importing the reference's own GE2ELoss on CPU.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

It imports ``embedding_model_GE2E/s3_loss_function_GE2E.py`` from
/root/reference (plain torch, no other deps), builds the minimal ``hp`` the
loss reads (``hp.general.small_err``, ``hp.general.device`` --
strings/constants.py:31,34), runs forward + ``loss.backward()`` in fp32 and in
fp64 on fixed inputs and writes inputs + outputs to ``tests/golden/*.npz``.
The fixtures hold data only; no reference source is stored.  The GPU box has
no /root/reference: tests there read only the .npz files.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, REF)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from embedding_model_GE2E.s3_loss_function_GE2E import GE2ELoss  # noqa: E402
from utils.dict_to_dot import GetDictWithDotNotation  # noqa: E402

from oracle.ge2e_oracle import synth_embeddings  # noqa: E402  (input generator only)

HP = GetDictWithDotNotation({"general": {"small_err": 1e-6, "device": torch.device("cpu")}})


def run_reference(emb_np, w, b, dtype):
    """Module path s3:19-30 + autograd, plus the static-method outputs s5 uses."""
    mod = GE2ELoss(HP)
    with torch.no_grad():
        mod.w.fill_(w)
        mod.b.fill_(b)
    mod = mod.to(dtype)
    e = torch.tensor(emb_np, dtype=dtype, requires_grad=True)
    loss = mod(e)
    loss.backward()
    with torch.no_grad():
        cent = GE2ELoss.get_centroids(e)
        loo = GE2ELoss.get_utterance_centroids(e)          # s3:95-112
        cos = GE2ELoss.get_cos_sim(e, cent, HP)
        _, per = GE2ELoss.calc_loss(mod.w * cos + mod.b, HP)
    return {
        "loss": loss.detach().numpy(),
        "per": per.numpy(),
        "cos": cos.numpy(),
        "cent": cent.numpy(),
        "loo": loo.numpy(),
        "dE": e.grad.numpy(),
        "dw": mod.w.grad.numpy(),
        "db": mod.b.grad.numpy(),
    }


def emit(name, emb, w=10.0, b=-5.0, keep_dE64=True):
    emb = np.ascontiguousarray(emb, dtype=np.float32)
    r32 = run_reference(emb, w, b, torch.float32)
    r64 = run_reference(emb, w, b, torch.float64)
    out = {"E": emb, "w": np.float32(w), "b": np.float32(b)}
    for k, v in r32.items():
        out[k] = np.asarray(v, dtype=np.float32)
    for k, v in r64.items():
        if k == "dE" and not keep_dE64:
            continue
        if k == "loo":          # fp32 copy is enough for a linear map
            continue
        out[k + "64"] = np.asarray(v, dtype=np.float64)
    path = os.path.join(HERE, name + ".npz")
    np.savez(path, **out)
    print(f"{name:28s} shape={emb.shape} loss={float(r32['loss']):.6f} "
          f"dw={float(r32['dw']):.6f} db={float(r32['db']):.3e} "
          f"({os.path.getsize(path) / 1024:.0f} KiB)")


def main():
    # G1: the only known-answer input in the reference (s3:144-145), one-hot 3x2x3
    toy = np.array([[0, 1, 0], [0, 0, 1], [0, 1, 0], [0, 1, 0], [1, 0, 0], [1, 0, 0]],
                   dtype=np.float32).reshape(3, 2, 3)
    emit("g1_toy_w1_b0", toy, w=1.0, b=0.0)
    emit("g1_toy_w10_b-5", toy)

    # G2 / G3: BASELINE.json configs 1 and 2, same generator as BASELINE.md section 2
    torch.manual_seed(1234)
    e = torch.nn.functional.normalize(torch.randn(4, 5, 256), dim=-1).numpy()
    emit("g2_cfg1_n4_m5_d256", e)
    torch.manual_seed(1234)
    e = torch.nn.functional.normalize(torch.randn(64, 10, 256), dim=-1).numpy()
    emit("g3_cfg2_n64_m10_d256", e, keep_dE64=False)

    # G4: non-unit-norm rows (pins the norm-gradient terms)
    emit("g4_raw_n16_m6_d128", synth_embeddings((16, 6, 128), "raw", seed=4))
    # G5: clustered (peaked softmax)
    emit("g5_clustered_n16_m6_d128", synth_embeddings((16, 6, 128), "clustered", seed=5))
    # G6: w = -3 pins "w is never clamped" (s3:22 is a no-op)
    emit("g6_wneg3_n8_m4_d64", synth_embeddings((8, 4, 64), "unit", seed=6), w=-3.0, b=0.5)
    # G7: M = 2 (smallest legal M), odd sizes, D not a multiple of 4
    emit("g7_m2_n5_d37", synth_embeddings((5, 2, 37), "raw", seed=7))
    emit("g7_n1_m3_d8", synth_embeddings((1, 3, 8), "raw", seed=8))
    # G8: degenerate norms: one zero row and one row below the cosine eps (1e-8)
    e = synth_embeddings((4, 3, 32), "unit", seed=9)
    e[1, 2, :] = 0.0
    e[2, 0, :] *= 1e-9
    emit("g8_degenerate_n4_m3_d32", e)
    # G9: a mid shape with N not a multiple of anything convenient
    emit("g9_n23_m7_d192", synth_embeddings((23, 7, 192), "clustered", seed=10))


if __name__ == "__main__":
    main()
