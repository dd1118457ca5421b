"""Rank process of the multi-process GPU tests (tests/test_dp_gpu.py): `python tests/dp_worker.py MODE RANK WORLD PORT OUT`.

By default every rank uses GPU 0 and the gloo backend (one GPU per box; RCCL refuses two ranks on one device); with
MT_TEST_BACKEND=nccl (set by the tests when the box has >= WORLD GPUs) rank r takes GPU r over RCCL -- the code path
is the same either way: dp.GradReducer's async all-reduces from inside the backward,
segmented hipGraph capture, DistributedDataParallel's reducer hooks on the nn.Module bridge.
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from modaltune_amd import dp, synth  # noqa: E402
from modaltune_amd.config import ModelConfig  # noqa: E402

L, SEED, NGRIDS, STEPS = 150, 5, 32, 4


def _cfg():
    return ModelConfig(depth=3, interaction_indexes=((0, 0), (1, 1), (2, 2)), slide_ngrids=NGRIDS)


def _slide(rank, sizes):
    inp = synth.synth_inputs(L + 13 * rank, sizes, seed=900 + rank, grid=NGRIDS)
    return (torch.from_numpy(inp["x"]).cuda(), torch.from_numpy(inp["coords"]).cuda(), [torch.from_numpy(a).cuda() for a in inp["genes"]],
            torch.from_numpy(inp["text"]).cuda())


def trainstep(rank, world, out):
    """STEPS data-parallel steps (one slide per rank, eager visit -> capture -> replays); dumps the final flat weights."""
    from modaltune_amd.engine import Engine
    from modaltune_amd.trainer import TrainStep
    sizes = synth.toy_group_sizes()
    cfg = _cfg()
    eng = Engine(cfg, sizes, "cuda")
    eng.load_state_dict(synth.synth_state_dict(cfg, sizes, SEED))
    dp.broadcast_params_(eng.store.flat)
    ts = TrainStep(eng, lr=1e-3, capture_after=1)
    ts.set_projector(synth.projector_state(SEED))
    x, coords, genes, text = _slide(rank, sizes)
    losses = []
    for _ in range(STEPS):
        losses.append(float(ts.step_graphed(x, coords, genes, text)))
    sd = eng.store.state_dict()      # what a checkpoint would hold: waits for the sharded parameter all-gather of the last step
    torch.cuda.synchronize()
    assert all(torch.equal(sd[k], eng.store.tensors[k]) for k in sd)
    nseg = max(len(s) for s in ts._graphs) if ts._graphs else 0
    np.savez(out, flat=eng.store.flat.cpu().numpy(), losses=np.array(losses), replays=ts.graph_replays, nseg=nseg,
             steps=int(ts.step_dev), buckets=len(ts.reducer.buckets), early=int(ts.buckets_started_early),
             split=int(ts._pass_streams is not None), sharded=int(ts.reducer.sharded), backend=dist.get_backend())


def seqpar_calls(rank, world, out):
    """The two collectives of modaltune_amd.seqpar on the process group as it is (device tensors as they are on `nccl`): with one rank
    both are identities -- what is checked is that RCCL accepts the calls (flat fp16 views, split sizes) and returns the payload."""
    from modaltune_amd import seqpar
    sp = seqpar.SeqParallelAttention.__new__(seqpar.SeqParallelAttention)
    sp.dist, sp.group, sp.W, sp.rank = dist, None, world, rank
    g = torch.Generator(device="cuda").manual_seed(3 + rank)
    kv = torch.randn(2, 16 * 37 * 48, device="cuda", generator=g).half()
    gathered = sp._all_gather(kv)
    pay = torch.randn(world * 4096, device="cuda", generator=g).half()
    recv = sp._all_to_all(pay, [4096] * world)
    torch.cuda.synchronize()
    np.savez(out, gather_ok=int(torch.equal(gathered[rank], kv)), shape=np.array(gathered.shape),
             a2a_ok=int(torch.equal(recv[rank * 4096:(rank + 1) * 4096], pay[rank * 4096:(rank + 1) * 4096])), backend=dist.get_backend())


def ragged(rank, world, out):
    """Ragged data-parallel steps: every rank walks the bag lengths in its OWN order with a two-entry graph LRU, so in the same
    step one rank replays a captured geometry while another captures or runs eager -- the collectives must still pair up (same
    buckets, same order) and every rank must end with the same weights."""
    from modaltune_amd.engine import Engine
    from modaltune_amd.trainer import TrainStep
    sizes = synth.toy_group_sizes()
    cfg = _cfg()
    eng = Engine(cfg, sizes, "cuda")
    eng.load_state_dict(synth.synth_state_dict(cfg, sizes, SEED))
    dp.broadcast_params_(eng.store.flat)
    ts = TrainStep(eng, lr=1e-3, capture_after=1, graph_cache_size=2)
    ts.set_projector(synth.projector_state(SEED))
    lengths = [150, 97, 230, 150, 64, 97]
    slides = {}
    for Lx in set(lengths):
        inp = synth.synth_inputs(Lx, sizes, seed=700 + 31 * rank + Lx, grid=NGRIDS)
        slides[Lx] = (torch.from_numpy(inp["x"]).cuda(), torch.from_numpy(inp["coords"]).cuda(), [torch.from_numpy(a).cuda() for a in inp["genes"]],
                      torch.from_numpy(inp["text"]).cuda())
    losses, modes = [], []
    for i in range(18):
        Lx = lengths[(i + 2 * rank) % len(lengths)] if i % 5 != 4 or rank == 0 else lengths[(i * 3 + 1) % len(lengths)]
        before = ts.graph_replays
        losses.append(float(ts.step_graphed(*slides[Lx])))
        modes.append(int(ts.graph_replays > before))
    sd = eng.store.state_dict()
    torch.cuda.synchronize()
    np.savez(out, flat=eng.store.flat.cpu().numpy(), losses=np.array(losses), modes=np.array(modes), steps=int(ts.step_dev),
             sharded=int(ts.reducer.sharded), keys=len(sd))


def titan(rank, world, out):
    """The TITAN configuration under the data-parallel TrainStep (eager schedule: every slide has its own token count): bucketed
    collectives from inside the backward through the native ViT blocks, sharded last bucket; the ranks end bit-identical."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import titan_standin
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_titan_cpu import TITAN_JSON
    from modaltune_amd.titan import NativeBackbone, TitanEngine, titan_model_config
    from modaltune_amd.trainer import TrainStep
    sizes = synth.toy_group_sizes()
    vit = titan_standin.VisionTransformer()
    titan_standin.init_standin(vit, SEED)
    cfg = titan_model_config(TITAN_JSON, 3, False, len(sizes))
    eng = TitanEngine(cfg, sizes, NativeBackbone(vit, "cuda"), "cuda")
    eng.load_state_dict(synth.synth_state_dict(cfg, sizes, SEED))
    dp.broadcast_params_(eng.store.flat)
    ts = TrainStep(eng, lr=1e-3)
    ts.set_projector(synth.projector_state(SEED))
    losses = []
    for i in range(5):
        Lx = 220 + 40 * ((i + rank) % 3)
        inp = synth.synth_inputs_titan(Lx, sizes, seed=300 + 17 * rank + i, grid=24)
        losses.append(float(ts.step(torch.from_numpy(inp["x"]).cuda().reshape(Lx, -1), torch.from_numpy(inp["coords"]).cuda().reshape(Lx, 2),
                                    [torch.from_numpy(a).cuda() for a in inp["genes"]], torch.from_numpy(inp["text"]).cuda(), update=True)))
    eng.store.state_dict()
    torch.cuda.synchronize()
    np.savez(out, flat=eng.store.flat.cpu().numpy(), losses=np.array(losses), steps=int(ts.step_dev), sharded=int(ts.reducer.sharded),
             impl=eng.backbone.kind)


def ddp_module(rank, world, out):
    """The reference's own multi-GPU form (utils/base_trainer.py:205-211): DistributedDataParallel around the nn.Module,
    3 forward calls, loss.backward().  Dumps the local (unwrapped) gradients and the DDP-averaged ones."""
    from modaltune_amd.aggregators import Aggregator
    from oracle import modaltune_oracle as O
    sizes = synth.toy_group_sizes()
    groups = {i: ["g"] * n for i, n in enumerate(sizes)}
    from modaltune_amd.config import GIGAPATH_JSON
    model = Aggregator.create("longnetvit_gene_adapter", gene_group_defination=groups, multi_task=3,
                              **dict(GIGAPATH_JSON, depth=3, slide_ngrids=NGRIDS, interaction_indexes=[[0, 0], [1, 1], [2, 2]], dropout=0.0,
                                     drop_path_rate=0.0))
    cfg = model.cfg
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(cfg, sizes, SEED).items()}, strict=True)
    x, coords, genes, text = _slide(rank, sizes)
    gd = {i: g for i, g in enumerate(genes)}
    psd = {k: torch.from_numpy(v).cuda() for k, v in synth.projector_state(SEED).items()}
    target = O.projector_forward(text, psd)      # [4, 256]; distill_loss picks rows 0, 1, 3

    def loss_of(m, xs=None):
        xs = x if xs is None else xs
        logits = torch.cat([m(x=xs, coords=coords, genes=gd, clinical=[], task_token=torch.eye(3)[t].cuda()) for t in (0, 1, 2)], dim=0)
        return O.distill_loss(logits, target)

    model.train()
    loss_of(model).backward()
    names = [k for k, p in model.named_parameters() if p.requires_grad]
    params = dict(model.named_parameters())
    local = torch.cat([params[k].grad.reshape(-1) for k in names]).clone()
    for k in names:
        params[k].grad = None
    ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[torch.cuda.current_device()])
    loss_of(ddp).backward()
    torch.cuda.synchronize()
    avg = torch.cat([params[k].grad.reshape(-1) for k in names]).clone()
    # second iteration: DDP's "expected to have finished reduction" check passes only if every hook fired
    for k in names:
        params[k].grad = None
    loss_of(ddp).backward()
    torch.cuda.synchronize()
    avg2 = torch.cat([params[k].grad.reshape(-1) for k in names]).clone()
    # steady state: a new slide tensor per iteration (same geometry) -- from the sixth batched pass on the bridge replays hipGraphs
    # (module_graph.ModuleReplay) under DDP's reducer hooks
    for _ in range(8):
        for k in names:
            params[k].grad = None
        loss_of(ddp, x.clone()).backward()
    torch.cuda.synchronize()
    avg3 = torch.cat([params[k].grad.reshape(-1) for k in names]).clone()
    np.savez(out, local=local.cpu().numpy(), avg=avg.cpu().numpy(), avg2=avg2.cpu().numpy(), avg3=avg3.cpu().numpy(),
             replays=int(model._replay.replays))


if __name__ == "__main__":
    mode, rank, world, port, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", port
    if os.environ.get("MT_TEST_BACKEND", "gloo") == "nccl":
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    else:
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        {"trainstep": trainstep, "ddp_module": ddp_module, "ragged": ragged, "titan": titan, "seqpar_calls": seqpar_calls}[mode](rank, world, out)
    finally:
        dist.destroy_process_group()
