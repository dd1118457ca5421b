"""LongNet sequence parallelism (dilated_attention.py:61-111): W rank processes on GPU 0 over gloo run
modaltune_amd.seqpar.SeqParallelAttention (all-gather of the k | v slab, long branches on the group's rows with the plan's
qlimit, reduce-scatter of dK / dV) and are compared with the oracle's restatement of the reference's sequence-parallel
DilatedAttention -- itself pinned to the reference by tests/golden/unit_seqpar.npz -- on the same fp16-rounded inputs."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "seqpar_worker.py")
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(case, world, tmp_path, timeout=600, backend="gloo"):
    port = str(_free_port())
    outs = [str(tmp_path / f"{case}_{backend}_{r}.npz") for r in range(world)]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), MT_TEST_BACKEND=backend)
    procs = [subprocess.Popen([sys.executable, WORKER, case, str(r), str(world), port, outs[r]], env=env,
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


def rel(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.mark.parametrize("case", ["w2", "w4", "big2", "big4"])
def test_sequence_parallel_attention_matches_oracle(case, tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    _check(case, tmp_path, "gloo")


@pytest.mark.parametrize("case", ["w2", "big2", "big4"])
def test_sequence_parallel_attention_over_rccl(case, tmp_path):
    """(c) one rank per GPU over RCCL: all_gather_into_tensor of the k | v slab and the fp16 all_to_all_single of the dK / dV
    partials on device tensors.  Skips on boxes with fewer GPUs than ranks."""
    import seqpar_worker as SW
    W = SW.CASES[case][0]
    if torch.cuda.device_count() < W:
        pytest.skip(f"needs {W} GPUs for one rank per GPU over RCCL (found {torch.cuda.device_count()})")
    _check(case, tmp_path, "nccl")


def _check(case, tmp_path, backend):
    import seqpar_worker as SW
    from oracle import modaltune_oracle as O
    W, B, L, segs, ratios = SW.CASES[case]
    got = _run(case, W, tmp_path, backend=backend)
    q16, k16, v16, dy16 = SW.rounded(case)
    qs = [(q16[r].double() / SW.QK).requires_grad_(True) for r in range(W)]
    ks = [k16[r].double().requires_grad_(True) for r in range(W)]
    vs = [v16[r].double().requires_grad_(True) for r in range(W)]
    mixed = O.dilated_attention_core_sp(qs, ks, vs, segs, ratios)
    ys = [torch.nn.functional.layer_norm(m, (768,), None, None, 1e-5) for m in mixed]
    sum((ys[r] * dy16[r].double()).sum() for r in range(W)).backward()
    assert int(got[0]["ngroups"]) >= 1 and int(got[0]["nb_loc"]) >= 1          # both kinds of branch are exercised
    for r in range(W):
        M = B * L
        assert rel(got[r]["y"].reshape(B, L, 768), ys[r].detach()) < 6e-3
        d = torch.from_numpy(got[r]["dqkv"]).double().view(B, L, 3, 16, 48)
        # the q columns are the gradient with respect to the PRE-SCALED q' = QK q
        assert rel(d[:, :, 0] * SW.QK, qs[r].grad) < 2e-2
        assert rel(d[:, :, 1], ks[r].grad) < 2e-2
        assert rel(d[:, :, 2], vs[r].grad) < 2e-2
