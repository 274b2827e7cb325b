"""Two-rank runs of the PRODUCT data-parallel step: MlpTrainer.train_step (HIP phases A / B / C + the two all-reduces of
wild_visual_navigation_amd.distributed) and TraversabilityEstimator.train (collective train / skip decision).

* backend "gloo" with CUDA tensors: both ranks share cuda:0, so this runs on the 1-GPU test box and exercises the real HIP
  phases, their stream ordering against the collectives, ragged and EMPTY shards;
* backend "nccl" (= RCCL): one rank per GPU, skipped when fewer than two GPUs are visible (the driver's 8-GPU node runs it).
Pass criteria: the 2-rank loss trajectory equals the single-process oracle trajectory on the concatenated batch (which is
pinned to the reference's SimpleMLP + TraversabilityLoss + torch.optim.Adam by tests/golden/mlp_train.pt), and the replicas
stay bit-identical."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _trainer_worker(rank, world, port, backend, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank if backend == "nccl" else 0),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    from wild_visual_navigation_amd import distributed as D
    from wild_visual_navigation_amd.model import SimpleMLP
    from wild_visual_navigation_amd.traversability_estimator import MlpTrainer

    r, w, local = D.init_from_env(backend=backend)
    dev = torch.device(f"cuda:{local}")
    torch.cuda.set_device(dev)
    c = torch.load(os.path.join(ROOT, "tests", "golden", "mlp_train.pt"), weights_only=False)["graph_pt_D90"]
    x, y, yv = c["x"].to(dev), c["y"].to(dev), c["y_valid"].to(dev)
    model = SimpleMLP(90, [256, 32, 1], True)
    model.load_state_dict(c["sd0"])
    model.to(dev)
    tr = MlpTrainer(model)
    tr.comm_events = []
    traj = []
    for step in range(10):
        if step % 2 == 0:      # ragged: 3 + 2 "frames" of 20 rows
            f0, f1 = D.shard_range(5, rank, world)
            lo, hi = f0 * 20, f1 * 20
        else:                  # rank 1's shard is EMPTY: it must still take part in both collectives
            lo, hi = (0, 100) if rank == 0 else (100, 100)
        losses = tr.train_step(x[lo:hi], y[lo:hi], yv[lo:hi])
        traj.append(losses.cpu().tolist())
    assert len(tr.comm_events) == 20
    torch.cuda.synchronize()
    D.barrier()
    q.put((rank, traj, {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}))
    torch.distributed.destroy_process_group()


def _run(worker, backend, n_out=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=worker, args=(r, 2, port, backend, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=280) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def _check_trainer(res):
    (_, t0, sd0), (_, t1, sd1) = res
    assert t0 == t1, "ranks disagree on the losses"
    for k in sd0:
        assert (sd0[k] == sd1[k]).all(), f"replicas diverged on {k}"
    c = torch.load(os.path.join(ROOT, "tests", "golden", "mlp_train.pt"), weights_only=False)["graph_pt_D90"]
    ref = c["traj"]  # the REFERENCE's SimpleMLP + TraversabilityLoss + torch.optim.Adam on the whole batch
    got = torch.tensor(t0)
    assert torch.allclose(got, ref.double().float(), rtol=3e-4, atol=2e-6), (got - ref).abs().max()
    for k in sd0:
        assert torch.allclose(torch.from_numpy(sd0[k]), c["sd10"][k], atol=3e-5), k


@pytest.mark.timeout(400)
def test_product_trainer_two_ranks_gloo_on_one_gpu(dev):
    _check_trainer(_run(_trainer_worker, "gloo"))


@pytest.mark.timeout(400)
def test_product_trainer_two_ranks_rccl(dev):
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device)")
    _check_trainer(_run(_trainer_worker, "nccl"))


def _estimator_worker(rank, world, port, backend, q):
    """TraversabilityEstimator.train on two replicas where rank 1 becomes ready LATER than rank 0: nobody may step (and hang
    in a collective) until every rank has a batch."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from wild_visual_navigation_amd import distributed as D
    from wild_visual_navigation_amd.cfg import ExperimentParams
    from wild_visual_navigation_amd.traversability_estimator import MissionNode, TraversabilityEstimator

    D.init_from_env(backend=backend)
    dev = torch.device("cuda:0")
    p = ExperimentParams()
    te = TraversabilityEstimator(p, device=dev, min_samples_for_training=1)
    g = torch.Generator().manual_seed(7 + rank)
    out = []
    for it in range(6):
        if rank == 0 or it >= 3:    # rank 1 receives no nodes during the first three rounds
            n = MissionNode(timestamp=float(it), pose_base_in_world=torch.eye(4) * 1.0)
            n.pose_base_in_world[0, 3] = float(it)       # one metre apart: far enough to be kept by the graph
            n.features = torch.randn(12, 90, generator=g).to(dev)
            n.feature_segments = torch.randint(0, 12, (32, 32), generator=g).to(dev)
            te.add_mission_node(n)
            mask = torch.full((3, 32, 32), float("nan"))
            mask[:, 8:24, 8:24] = 0.8
            te.update_supervision(n, mask.to(dev))
        res = te.train()
        out.append(res.get("loss_total", None))
    D.barrier()
    q.put((rank, out, te.step))
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(400)
def test_estimator_train_skips_collectively(dev):
    (_, l0, s0), (_, l1, s1) = _run(_estimator_worker, "gloo")
    assert s0 == s1, "replicas took a different number of optimisation steps"
    assert s0 >= 1
    stepped0 = [v is not None and v != -1 for v in l0]
    stepped1 = [v is not None and v != -1 for v in l1]
    assert stepped0 == stepped1
    assert [a for a, ok in zip(l0, stepped0) if ok] == [a for a, ok in zip(l1, stepped1) if ok]


def test_bench_two_ranks_through_its_own_launcher(tmp_path):
    """The driver's 8-GPU tier is the first time bench.py meets torch.distributed.run: run that exact path here with two ranks
    (bench.py re-executes itself under torch.distributed.run; --backend gloo lets both ranks share the box's one GPU).  Checks
    the launcher, the rendezvous, ranks_seen, the per-step all-reduce timings, and that n_gpus / value follow the rank count."""
    import json
    import subprocess

    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
           "--backend", "gloo", "--batch", "8", "--chunk", "8"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                          # rank 0 alone prints, exactly one line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["ranks_seen"] == 2 and out["steps"] == 3 and out["warmup"] == 1
    assert out["scaling"] == "weak" and out["config"]["frames_per_gpu_per_step"] == 8
    assert abs(out["value"] - 2 * 8 * 3 / (out["ms_per_step"] * 3e-3)) / out["value"] < 0.02      # whole-job frames / max-over-ranks time
    ar = out["allreduce_ms"]
    assert ar["per_step_total"] > 0 and ar["stats_allreduce"]["median"] > 0 and ar["grad_allreduce"]["median"] > 0
    assert out["replicas_identical_after_timed_steps"] is True      # parameters, Adam moments and losses bit-identical across the ranks
    assert "parity_mode" not in out and "stego_fast" not in out     # the extra legs are an N = 1 matter
    assert out["roofline"]["achieved"] > 0 and out["final_loss"] == out["final_loss"]
