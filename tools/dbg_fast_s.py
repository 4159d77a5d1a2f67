import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from oracle import ge2e_oracle as orc
from speaker_embedding_ge2e_loss_amd import functional as GF
from test_gpu_team_fwd import run_fwd
for kind in ("clustered", "unit"):
    E = orc.synth_embeddings((37, 64, 10, 256), kind, seed=7)
    ref = orc.closed_form(E, 10.0, -5.0)
    o1 = run_fwd(GF, E, 10.0, -5.0, "softmax", want_wb=True)
    o0 = run_fwd(GF, E, 10.0, -5.0, "softmax", want_wb=False)
    for nm, o in (("general", o1), ("fast", o0)):
        dl = np.abs(o["loss"] - ref["loss"]) / np.abs(ref["loss"])
        dp = np.abs(o["per"] - ref["per"])
        print(kind, nm, "loss rel max %.3e" % dl.max(), "per abs max %.3e" % dp.max(), "per rel max %.3e" % (dp / np.maximum(np.abs(ref["per"]), 1e-30)).max(), "loss[0]", o["loss"][0], ref["loss"][0])
    i = np.unravel_index(np.argmax(np.abs(o0["per"] - o1["per"])), o0["per"].shape)
    print("worst fast-vs-general per at", i, o0["per"][i], o1["per"][i], ref["per"][i])
