"""Multi-process GPU tests of the data-parallel path: 2 fresh rank processes on GPU 0 over gloo.

(a) TrainStep.step_graphed on a different slide per rank: eager visit, segmented hipGraph capture (the backward is cut
    where a gradient bucket becomes final), replays; both ranks end with identical weights, equal to a single-process
    run that averages the two slides' gradients.
(b) the nn.Module bridge wrapped in DistributedDataParallel (utils/base_trainer.py:205-211): its reducer hooks fire and
    param.grad is the mean of the per-rank gradients; a second iteration passes DDP's "finished reduction" check.
(c) `bench.py --gpus 2 --backend gloo` launches its two ranks itself and prints n_gpus: 2.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "dp_worker.py")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_ranks(mode, tmp_path, world=2, timeout=600, backend="gloo", extra_env=None):
    port = str(_free_port())
    outs = [str(tmp_path / f"{mode}_{backend}_{r}.npz") for r in range(world)]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), MT_TEST_BACKEND=backend)
    env.update(extra_env or {})
    procs = [subprocess.Popen([sys.executable, WORKER, mode, str(r), str(world), port, outs[r]], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o.decode()[-3000:])
    assert all(p.returncode == 0 for p in procs), "\n----\n".join(logs)
    return [np.load(o) for o in outs]


def _need_gpus(n):
    """RCCL tests: one rank per GPU.  They skip cleanly on the 1-GPU boxes and run the moment a box has n GPUs."""
    if torch.cuda.device_count() < n:
        pytest.skip(f"needs {n} GPUs for one rank per GPU over RCCL (found {torch.cuda.device_count()})")


def test_two_rank_trainstep_matches_gradient_averaging(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    _check_trainstep(_run_ranks("trainstep", tmp_path))


def test_two_rank_trainstep_with_pass_groups_on_two_streams(tmp_path):
    """The same with the task passes of every rank's step split into two concurrent groups (forced on at test size).  Round 6: the
    groups' backwards run stage by stage and every bucket but the last is summed over the two gradient sets and handed to the
    reducer as soon as BOTH groups have left its interaction block (`dp_schedule = "groups_joined"`, the default): >= 3 of the 4
    buckets start before the last backward kernel is enqueued, the capture is cut at those joins (4 segments), and the ranks end
    bit-identical and equal to the single-process reference."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    res = _run_ranks("trainstep", tmp_path, extra_env={"MT_SPLIT_PASSES": "force"})
    assert all(int(r["split"]) == 1 and int(r["early"]) >= 3 for r in res), [(int(r["split"]), int(r["early"])) for r in res]
    _check_trainstep(res, nseg=4)


def test_two_rank_trainstep_with_pass_groups_and_exposed_reduction(tmp_path):
    """Round 5's form of the above, still selectable (`MT_DP_SCHEDULE=groups_exposed`; bench.py --gpus N times it against the other two
    schedules): no bucket starts before the groups' streams have met (ONE captured segment), the collectives run behind the join."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    res = _run_ranks("trainstep", tmp_path, extra_env={"MT_SPLIT_PASSES": "force", "MT_DP_SCHEDULE": "groups_exposed"})
    assert all(int(r["split"]) == 1 and int(r["early"]) == 0 for r in res)
    _check_trainstep(res, nseg=1)


def test_two_rank_trainstep_with_the_batched_schedule_forced(tmp_path):
    """`MT_DP_SCHEDULE=batched`: one B = 3 pass even where the pass groups would run (buckets started from inside its backward)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    res = _run_ranks("trainstep", tmp_path, extra_env={"MT_SPLIT_PASSES": "force", "MT_DP_SCHEDULE": "batched"})
    assert all(int(r["split"]) == 0 for r in res)
    _check_trainstep(res, nseg=4)


@pytest.mark.parametrize("sched,nseg,split,early", [("groups_joined", 4, 1, 3), ("groups_exposed", 1, 1, 0), ("batched", 4, 0, 0)])
def test_one_rank_rccl_rehearsal_issues_every_collective_of_the_step(tmp_path, sched, nseg, split, early):
    """RCCL on a one-GPU box: ONE rank on the `nccl` backend with MT_DP_REHEARSE=1 (dp.single_rank_rehearsal) -- the constructor
    broadcast, every bucket's asynchronous all-reduce from inside the (segmented, captured) backward, the sharded bucket's
    reduce-scatter / AdamW on the shard / all-gather of the parameters and the found_inf MAX all run over a one-rank RCCL
    communicator, under each of the three data-parallel schedules.  The sum over one rank is the identity, so the run must land
    on the plain single-process training run of the same slide."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    (r,) = _run_ranks("trainstep", tmp_path, world=1, backend="nccl",
                      extra_env={"MT_DP_REHEARSE": "1", "MT_SPLIT_PASSES": "force", "MT_DP_SCHEDULE": sched})
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dp_worker as W
    from modaltune_amd import synth
    from modaltune_amd.engine import Engine
    from modaltune_amd.trainer import TrainStep
    assert str(r["backend"]) == "nccl" and int(r["sharded"]) == 1 and int(r["buckets"]) == 4
    assert int(r["steps"]) == W.STEPS and int(r["replays"]) == W.STEPS - 1 and int(r["nseg"]) == nseg
    assert int(r["split"]) == split and int(r["early"]) >= early
    sizes = synth.toy_group_sizes()
    cfg = W._cfg()
    eng = Engine(cfg, sizes, "cuda")
    eng.load_state_dict(synth.synth_state_dict(cfg, sizes, W.SEED))
    ts = TrainStep(eng, lr=1e-3)
    ts.set_projector(synth.projector_state(W.SEED))
    slide = W._slide(0, sizes)
    ref_losses = np.array([float(ts.step(*slide)) for _ in range(W.STEPS)])
    torch.cuda.synchronize()
    assert np.allclose(r["losses"], ref_losses, rtol=2e-3), (r["losses"], ref_losses)
    d = np.abs(r["flat"] - eng.store.flat.cpu().numpy())
    assert d.max() <= 2.0 * W.STEPS * 1e-3 + 1e-7                    # (same bars as the two-rank check below)
    assert np.mean(d > 0.05 * 1e-3) < 0.02, (np.mean(d > 0.05 * 1e-3), d.max())


def test_one_rank_rccl_rehearsal_of_the_sequence_parallel_collectives(tmp_path):
    """seqpar's all_gather_into_tensor / all_to_all_single on device tensors over a one-rank RCCL communicator (the `else:` arms that
    gloo's host staging never takes)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    (r,) = _run_ranks("seqpar_calls", tmp_path, world=1, backend="nccl")
    assert str(r["backend"]) == "nccl" and int(r["gather_ok"]) == 1 and int(r["a2a_ok"]) == 1 and tuple(r["shape"]) == (1, 2, 16 * 37 * 48)


def test_bench_one_rank_rccl_rehearsal():
    """`bench.py --dp-rehearsal`: the data-parallel bench path (schedule candidates, comm record) on a one-rank RCCL communicator."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dp-rehearsal", "--patches", "7600", "--steps", "3", "--warmup", "3",
                        "--no-cpu-baseline"], env=env, capture_output=True, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    out = json.loads([l for l in p.stdout.decode().splitlines() if l.startswith("{")][-1])
    c = out["comm"]
    assert out["n_gpus"] == 1 and c["backend"] == "nccl" and c["ranks_seen_by_all_reduce"] == 1 and c["last_bucket_sharded"] is True
    assert set(c["schedule_timings"]) == {"groups_joined", "groups_exposed", "batched"}
    assert all("ms_per_step" in v for v in c["schedule_timings"].values()), c["schedule_timings"]
    assert out["launch"] == "hipGraph replay" and out["skipped_steps"] == 0 and "dp_rehearsal" in out["config"]


def _check_ragged(res):
    assert all(int(r["sharded"]) == 1 and int(r["steps"]) == 18 for r in res)
    assert np.isfinite(res[0]["losses"]).all() and np.isfinite(res[1]["losses"]).all()
    assert (res[0]["modes"] != res[1]["modes"]).any(), "the ranks never disagreed on replay vs capture / eager: the test lost its point"
    assert np.array_equal(res[0]["flat"], res[1]["flat"])          # same averaged gradients, same updates: bit-identical weights


def test_two_rank_ragged_steps_with_rank_local_capture_decisions(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    _check_ragged(_run_ranks("ragged", tmp_path))


def test_two_rank_ragged_steps_over_rccl(tmp_path):
    _need_gpus(2)
    _check_ragged(_run_ranks("ragged", tmp_path, backend="nccl"))


def test_two_rank_titan_trainstep(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    res = _run_ranks("titan", tmp_path)
    assert all(int(r["sharded"]) == 1 and int(r["steps"]) == 5 and str(r["impl"]) == "native" for r in res)
    assert np.isfinite(res[0]["losses"]).all() and np.isfinite(res[1]["losses"]).all()
    assert np.array_equal(res[0]["flat"], res[1]["flat"])
    assert not np.array_equal(res[0]["losses"], res[1]["losses"])      # (different slides per rank)


def test_two_rank_trainstep_over_rccl(tmp_path):
    """(a) over RCCL, one rank per GPU: GradReducer's async collectives between segmented hipGraph replays on RCCL's stream."""
    _need_gpus(2)
    _check_trainstep(_run_ranks("trainstep", tmp_path, backend="nccl"))


def _check_trainstep(results, nseg=4):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dp_worker as W
    from modaltune_amd import synth
    from modaltune_amd.engine import Engine
    from modaltune_amd.trainer import TrainStep
    r0, r1 = results
    assert int(r0["steps"]) == W.STEPS and int(r0["replays"]) == W.STEPS - 1      # one eager visit, then capture + replays
    assert int(r0["buckets"]) == 4 and int(r0["nseg"]) == nseg                     # (4: the backward was cut at every bucket boundary)
    assert np.array_equal(r0["flat"], r1["flat"])                                  # same reduced gradient -> same weights, bitwise
    assert not np.allclose(r0["losses"], r1["losses"])                             # (different slides)
    # single-process reference: gradients of both slides summed, AdamW with grad_mult = 1/2
    sizes = synth.toy_group_sizes()
    cfg = W._cfg()
    eng = Engine(cfg, sizes, "cuda")
    eng.load_state_dict(synth.synth_state_dict(cfg, sizes, W.SEED))
    ts = TrainStep(eng, lr=1e-3)
    ts.set_projector(synth.projector_state(W.SEED))
    slides = [W._slide(r, sizes) for r in range(2)]
    ref_losses = []
    for _ in range(W.STEPS):
        acc = torch.zeros_like(eng.store.flat_grad)
        ls = []
        for x, coords, genes, text in slides:
            ls.append(float(ts.step(x, coords, genes, text, update=False)))
            acc += eng.store.flat_grad
        eng.store.flat_grad.copy_(acc)
        ts._adam_and_refresh(2)
        ref_losses.append(ls)
    torch.cuda.synchronize()
    ref_losses = np.array(ref_losses)
    assert np.allclose(r0["losses"], ref_losses[:, 0], rtol=2e-3), (r0["losses"], ref_losses[:, 0])
    assert np.allclose(r1["losses"], ref_losses[:, 1], rtol=2e-3), (r1["losses"], ref_losses[:, 1])
    ref = eng.store.flat.cpu().numpy()
    moved = np.abs(ref - synth_flat(eng, cfg, sizes, W.SEED))
    d = np.abs(r0["flat"] - ref)
    # AdamW's normalised update is ~lr per step whatever the gradient's size: elements whose gradient is rounding noise may
    # differ by up to 2 lr per step between two runs (fp32 atomics reorder); everything else agrees to a small fraction of lr
    assert d.max() <= 2.0 * W.STEPS * 1e-3 + 1e-7
    assert np.mean(d > 0.05 * 1e-3) < 0.02, (np.mean(d > 0.05 * 1e-3), d.max())
    assert moved.mean() > 0.5e-3                                                   # (the weights did move)


def synth_flat(eng, cfg, sizes, seed):
    """The initial flat trainable buffer (state_dict order, 16-byte aligned slots)."""
    from modaltune_amd import synth
    sd = synth.synth_state_dict(cfg, sizes, seed)
    out = np.zeros(eng.store.n_flat, dtype=np.float32)
    for k, (o, n, _) in eng.store.slots.items():
        out[o:o + n] = sd[k].reshape(-1)
    return out


def test_module_under_distributed_data_parallel(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    _check_ddp(_run_ranks("ddp_module", tmp_path))


def test_module_under_distributed_data_parallel_over_rccl(tmp_path):
    """(b) over RCCL: DistributedDataParallel(device_ids=[rank]) around the nn.Module bridge, one rank per GPU."""
    _need_gpus(2)
    _check_ddp(_run_ranks("ddp_module", tmp_path, backend="nccl"))


def _check_ddp(results):
    r0, r1 = results
    mean = 0.5 * (r0["local"].astype(np.float64) + r1["local"].astype(np.float64))
    scale = np.abs(mean).max()
    for r in (r0, r1):
        assert np.abs(r["avg"] - mean).max() < 2e-3 * scale        # DDP averaged exactly what the flat all-reduce path sums
        assert np.abs(r["avg2"] - mean).max() < 2e-3 * scale       # and a second iteration reduces again
        assert int(r["replays"]) >= 2 and np.abs(r["avg3"] - mean).max() < 2e-3 * scale      # ... and so do the bridge's graph replays
    assert np.array_equal(r0["avg"], r1["avg"])
    assert np.abs(r0["local"] - r1["local"]).max() > 1e-3 * scale  # (the ranks saw different slides)


def test_bench_launches_its_own_ranks():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    _check_bench("gloo")


def test_bench_two_gpus_over_rccl():
    """(d) `bench.py --gpus 2` over RCCL: fresh rank processes, one per GPU; the step must run as hipGraph replays (a capture
    failure with world > 1 exits non-zero instead of silently timing eager launches)."""
    _need_gpus(2)
    _check_bench("nccl")


def test_bench_two_ranks_choose_the_data_parallel_schedule_of_a_long_bag():
    """Round 6: at a bag length where the pass groups run, `bench.py --gpus N` captures and times the three
    data-parallel schedules and keeps the fastest; the line carries the choice, the three timings and -- when the joined schedule is
    chosen -- how many buckets were started before the backward had ended."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    out = _check_bench("gloo", patches="7600")
    c = out["comm"]
    assert set(c["schedule_timings"]) == {"groups_joined", "groups_exposed", "batched"} and c["schedule_chosen"] in c["schedule_timings"]
    assert all(v["ms_per_step"] > 0 for v in c["schedule_timings"].values())
    assert c["schedule_timings"][c["schedule_chosen"]]["ms_per_step"] == min(v["ms_per_step"] for v in c["schedule_timings"].values())
    if c["schedule_chosen"] == "groups_joined":
        assert c["buckets_started_before_the_backward_ended"] >= 3


def _check_bench(backend, patches="1024"):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", backend, "--patches", patches,
                        "--steps", "3", "--warmup", "3", "--no-cpu-baseline"], env=env, capture_output=True, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    line = [l for l in p.stdout.decode().splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "dp2" and out["scaling"] == "weak"
    assert out["launch"] == "hipGraph replay" and out["graph_replays"] >= 3
    assert out["value"] > 0 and out["skipped_steps"] == 0
    # the N > 1 line explains itself -- the collective really ran over `world` ranks, the bucket sizes, and how
    # much of the gradient / parameter exchange was EXPOSED on the compute stream (HIP events, max over ranks)
    c = out["comm"]
    assert c["ranks_seen_by_all_reduce"] == 2 and c["backend"] == backend and c["last_bucket_sharded"] is True
    assert len(c["bucket_bytes"]) == 4 and all(b > 0 for b in c["bucket_bytes"]) and sum(c["bucket_bytes"]) > 30e6
    assert c["comm_exposed_events"] == 3 and c["comm_exposed_ms"] >= 0.0
    assert c["param_gather_exposed_ms"] >= 0.0 and len(c["per_rank_ms_per_step"]) == 2
    return out
    assert max(c["per_rank_ms_per_step"]) <= out["ms_per_step"] * 1.001
