"""`python bench.py --gpus N` outside a launcher must start N ranks itself (as a child process, before any GPU call) and
rank 0 must print one JSON line with n_gpus = N.  Device-less dry run: gloo rendezvous on 127.0.0.1, the real
1,464,578-float flat-bucket all-reduce of --mode train-step on CPU tensors, no kernel launched (value is null)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(*extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env["OMP_NUM_THREADS"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--steps", "2", "--warmup", "1", *extra],
                       capture_output=True, text=True, timeout=240, env=env, cwd="/tmp")
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout  # ONE json line, from rank 0 only
    return json.loads(lines[0])


def test_gpus_2_spawns_two_ranks_train_step():
    j = run("--gpus", "2", "--mode", "train-step", "--config", "cfg4")
    assert j["n_gpus"] == 2 and j["dry_run"] is True and j["value"] is None
    assert j["config"]["N"] == 256 and j["config"]["mode"] == "train-step" and j["config"]["batches_per_launch"] == 1
    t = j["train_step"]
    assert t["bucket_floats"] == 1464578 and t["bucket_bytes"] == 4 * 1464578
    assert t["ms_with_allreduce"] > t["ms_without_allreduce"] >= 0 and t["allreduce_alone_ms"] > 0
    assert "dp2" in j["config"]["parallelism"]


def test_gpus_8_cfg4_train_step_dry_run():
    """The configuration BASELINE names for the 8-GPU run (cfg4, data parallel, one flat-bucket all-reduce per step) with
    eight ranks on CPU: the backend counts eight ranks in a real collective and every rank's clock reaches the line."""
    j = run("--gpus", "8", "--mode", "train-step", "--config", "cfg4")
    assert j["n_gpus"] == 8 and j["rccl_ranks"] == 8 and len(j["per_rank_value"]) == 8
    assert j["config"]["N"] == 256 and "dp8" in j["config"]["parallelism"]
    assert j["train_step"]["allreduce_busbw_GBs"] > 0


def test_gpus_8_cfg2_loss_mode_dry_run():
    """The driver's scaling command (`--gpus 8`, default mode and config = the metric line) with eight ranks on CPU: the
    line names cfg2's workload, weak scaling over whole batches, eight ranks counted by a real collective."""
    j = run("--gpus", "8", "--config", "cfg2")
    assert j["n_gpus"] == 8 and j["rccl_ranks"] == 8 and len(j["per_rank_value"]) == 8
    assert j["config"]["mode"] == "loss" and j["config"]["N"] == 64 and j["config"]["batches_per_launch"] == 16384
    assert j["scaling"] == "weak" and "dp8" in j["config"]["parallelism"] and "train_step" not in j
    assert j["metric"].startswith("GE2E loss+backward throughput") and j["unit"] == "batches/s"


def test_synth_is_generated_on_the_device_it_is_asked_for():
    """bench.synth builds the embeddings where they will live (no host-side 10 GB block per rank): unit rows, seeded."""
    import importlib.util
    import torch
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    a = bench.synth(5, 3, 4, 64, 7, "cpu")
    b = bench.synth(5, 3, 4, 64, 7, "cpu")
    assert a.shape == (5, 3, 4, 64) and a.dtype == torch.float32 and torch.equal(a, b)
    assert torch.allclose(a.norm(dim=-1), torch.ones(5, 3, 4), atol=1e-6)
    assert not torch.equal(a, bench.synth(5, 3, 4, 64, 8, "cpu"))


def test_gpus_1_stays_one_process():
    j = run("--gpus", "1")
    assert j["n_gpus"] == 1 and j["config"]["batches_per_launch"] == 16384 and j["config"]["mode"] == "loss"
    assert j["rccl_ranks"] == 1 and len(j["per_rank_value"]) == 1


def test_bucket_is_the_reference_encoder_plus_w_b():
    from speaker_embedding_ge2e_loss_amd.encoder import SpeakerEncoder
    enc = SpeakerEncoder()  # 80 mels, 256 hidden, 3 layers, 256-d embedding: strings/constants.py:52,90-92
    assert sum(p.numel() for p in enc.parameters()) + 2 == 1464578
