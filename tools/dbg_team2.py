#!/usr/bin/env python3
"""Development aid: a -DGE2E_T2_DEBUG build dumps, after GE, everything GE read (e-hat images, dL/dS images, row
coefficients, the centroid fragments in registers) and what it produced (held).  The host recomputes every held tile
from the dumped inputs and reports how the tiles that are wrong in the final dE deviate."""
import ctypes as C
import os
import subprocess
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import ge2e_oracle as orc  # noqa: E402
from speaker_embedding_ge2e_loss_amd import build  # noqa: E402

lib_path = os.path.join(build.PKG_DIR, "libge2e_hip_exp_dbg.so")
build.build_variant(lib_path, ["-DGE2E_T2_DEBUG"])
os.environ["GE2E_HIP_LIB"] = lib_path
from speaker_embedding_ge2e_loss_amd import _lib, functional as GF  # noqa: E402

lib = _lib.load()
raw = C.CDLL(lib_path)
shape = (32, 64, 10, 256)
B, N, M, D = shape
P, GP, RT = D + 16, 72, 80
E = orc.synth_embeddings(shape, "unit", seed=sum(shape))
ref = orc.closed_form(E, 10.0, -5.0)
dev = torch.device("cuda:0")
e = torch.as_tensor(E, device=dev)
w, b = torch.tensor(10.0, device=dev), torch.tensor(-5.0, device=dev)
for rep in range(2):
    dump = torch.zeros(B * 8 * 131072, dtype=torch.int32, device=dev)
    raw.ge2e_debug_set_t2_dump(C.c_void_p(dump.data_ptr()))
    o = GF.loss_fwd_bwd(e, w, b, impl="team2")
    torch.cuda.synchronize()
    dE = o.dE.cpu().numpy().reshape(B, 8, 5, 16, 16, 16)       # batch, member, rb, row, tile, col
    rf = ref["dE"].reshape(B, 8, 5, 16, 16, 16)
    err = np.abs(dE - rf).max(axis=(3, 5)) / np.abs(rf).max()   # batch, member, rb, tile
    bad = np.argwhere(err > 1e-4)
    dmp = dump.cpu().numpy().reshape(B, 8, 131072)
    print("rep", rep, "bad tiles", len(bad), flush=True)
    for bi, m, rb, tile in bad[:3]:
        wave, i = tile % 8, tile // 8
        d = dmp[bi, m]
        et = d[:2 * RT * P // 2].view(np.float16).reshape(2, RT, P).astype(np.float64)      # hi, lo images
        g = d[22000:22000 + 2 * RT * GP // 2].view(np.float16).reshape(2, RT, GP).astype(np.float64)
        rs = d[28000:28000 + RT * 8].view(np.float32).reshape(RT, 8).astype(np.float64)
        ga = d[30000:30000 + 8 * 512 * 4].reshape(2, 2, 2, 512, 4)[i, :, :, 64 * wave:64 * wave + 64]  # s2, hl, lane, 4 dwords
        gaf = ga.copy().view(np.float16).reshape(2, 2, 64, 8).astype(np.float64)          # s2, hl, lane, 8 halfs
        held = d[50000:50000 + 10 * 512 * 4].view(np.float32).reshape(2, 5, 512, 4)[i, rb, 64 * wave:64 * wave + 64].astype(np.float64)
        # CH^T fragment: lane (l15, q) holds column d = 16 dt + l15, slots 32 s2 + 8 q + 0..7
        cht = np.zeros((2, 16, 64))                                                        # hl, d_local, slot
        for lane in range(64):
            l15, q = lane & 15, lane >> 4
            for s2 in range(2):
                for hl in range(2):
                    cht[hl, l15, 32 * s2 + 8 * q:32 * s2 + 8 * q + 8] = gaf[s2, hl, lane]
        rows = slice(16 * rb, 16 * rb + 16)
        gh, gl = g[0, rows, :64], g[1, rows, :64]
        acc = gh @ cht[0].T + gh @ cht[1].T + gl @ cht[0].T                              # [row][d_local], scaled 2^16
        ra, c1 = rs[rows, 4], rs[rows, 5]
        dt = wave + 8 * i
        eh, el = et[0, rows, 16 * dt:16 * dt + 16], et[1, rows, 16 * dt:16 * dt + 16]
        want = acc * ra[:, None] + c1[:, None] * (eh + el)                                # [row][d_local]
        got = np.zeros((16, 16))
        for lane in range(64):
            l15, q = lane & 15, lane >> 4
            got[l15, 4 * q:4 * q + 4] = held[lane]
        dev_ = got - want
        t_acc = acc * ra[:, None]
        t_e = c1[:, None] * (eh + el)
        print(f"  batch {bi} member {m} rb {rb} tile {tile} (wave {wave}, i {i}): |got-want| max {np.abs(dev_).max():.3e} "
              f"(|want| max {np.abs(want).max():.3e}); ratio dev/t_acc median {np.median(dev_ / (t_acc + 1e-30)):.3f}, "
              f"dev/t_e median {np.median(dev_ / (t_e + 1e-30)):.3f}; rows with error {np.nonzero(np.abs(dev_).max(axis=1) > 1e-6)[0].tolist()}, "
              f"cols {np.nonzero(np.abs(dev_).max(axis=0) > 1e-6)[0].tolist()}")
        cc = int(np.argmax(np.abs(dev_).max(axis=0)))
        np.set_printoptions(precision=4, linewidth=200, suppress=True)
        print("     col", cc, "got   ", got[:6, cc]); print("            want  ", want[:6, cc]); print("            t_acc ", t_acc[:6, cc]); print("            t_e   ", t_e[:6, cc])
        print("            got/ra", (got[:6, cc] / ra[:6]), " acc", acc[:6, cc], " got-t_e", (got - t_e)[:6, cc], "ra", ra[:3], "c1", c1[:3])
        # does `got` match the value of another row block or tile?
        for rb2 in range(5):
            rows2 = slice(16 * rb2, 16 * rb2 + 16)
            acc2 = g[0, rows2, :64] @ cht[0].T + g[0, rows2, :64] @ cht[1].T + g[1, rows2, :64] @ cht[0].T
            for dt2 in (dt,):
                w2 = acc2 * rs[rows2, 4][:, None] + rs[rows2, 5][:, None] * (et[0, rows2, 16 * dt2:16 * dt2 + 16] + et[1, rows2, 16 * dt2:16 * dt2 + 16])
                if np.abs(got - w2).max() < 1e-5:
                    print(f"     -> equals the value of row block {rb2}")
            mix = acc * ra[:, None] + rs[rows2, 5][:, None] * (et[0, rows2, 16 * dt:16 * dt + 16] + et[1, rows2, 16 * dt:16 * dt + 16])
            if rb2 != rb and np.abs(got - mix).max() < 1e-5:
                print(f"     -> own accumulator, but e-hat / c1 of row block {rb2}")
            mix2 = acc2 * rs[rows2, 4][:, None] + c1[:, None] * (eh + el)
            if rb2 != rb and np.abs(got - mix2).max() < 1e-5:
                print(f"     -> own e-hat term, but accumulator / ra of row block {rb2}")
