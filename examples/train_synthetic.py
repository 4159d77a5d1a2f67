"""The reference's train -> test-loss -> EER flow (train_embedding_model.py / test_embedding_model.py) on synthetic
speakers, with every hot piece on the GPU.  One process per GPU:

    python examples/train_synthetic.py                       # one GPU
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/train_synthetic.py

s1 (loader)  -> data.SpectrogramStore + GE2EBatchSampler   (resident arrays, one gather-and-cast launch per batch)
s2 (encoder) -> encoder.SpeakerEncoder(normalize=False)     (LSTM + Linear through torch; the tail is fused below)
s3 (loss)    -> GE2ELoss                                     (one HIP launch: loss + every gradient)
s4 (trainer) -> trainer.DPTrainer(fused_tail=True)           (perm / un-perm, clips, SGD, flat-bucket RCCL all-reduce)
s5 (EER)     -> evaluation.calculate_ERR                     (HIP cosines + threshold-sweep counts)
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from speaker_embedding_ge2e_loss_amd import GE2ELoss, HParams  # noqa: E402
from speaker_embedding_ge2e_loss_amd.data import GE2EBatchSampler, SpectrogramStore  # noqa: E402
from speaker_embedding_ge2e_loss_amd.encoder import SpeakerEncoder  # noqa: E402
from speaker_embedding_ge2e_loss_amd.evaluation import calculate_ERR  # noqa: E402
from speaker_embedding_ge2e_loss_amd.trainer import DPTrainer  # noqa: E402


def main():
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)))
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    # synthetic "sv_<speaker>.npy" arrays: (utterances, frames, mels) float64, a spectral signature per speaker
    S, U, T, F, N, M = 64, 20, 180, 80, 8, 10
    rng = np.random.default_rng(0)
    sig = rng.standard_normal((S, 1, 1, F))
    store = SpectrogramStore([sig[j] + 0.5 * rng.standard_normal((U, T, F)) for j in range(S)], device=dev)
    np.random.seed(100 + rank)                    # every rank draws its own utterances and crops
    sampler = GE2EBatchSampler(store, utter_num=M, min_utter_len=160, training=True)

    torch.manual_seed(0)
    hp = HParams(device=dev)
    encoder = SpeakerEncoder(F, 256, 3, 256, normalize=False).to(dev)   # the reference's sizes (strings/constants.py)
    trainer = DPTrainer(encoder, GE2ELoss(hp), lr=0.05, seed=rank, fused_tail=True)

    gen = torch.Generator().manual_seed(rank)
    for epoch in range(3):
        losses = [trainer.step(mel) for mel in sampler.loader(batch_size=N, generator=gen)]
        mean = float(torch.stack(losses).mean())
        test = trainer.eval_loss(list(sampler.loader(batch_size=N, shuffle=False))[:2])
        if rank == 0:
            print(f"epoch {epoch}: train loss {mean:.3f}  test loss {test:.3f}", flush=True)

    if rank == 0:
        eval_encoder = SpeakerEncoder(F, 256, 3, 256, normalize=True).to(dev)
        eval_encoder.load_state_dict(encoder.state_dict())
        batch = next(iter(sampler.loader(batch_size=4, shuffle=False)))        # (4, M, frames, mels)
        calculate_ERR(eval_encoder.eval(), hp, N=4, M=M, test_loader=[batch])
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
