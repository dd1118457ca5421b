"""Do the task passes overlap better as TWO concurrent engine passes (B = 2 and B = 1 on two HIP streams: one group's HBM-bound
LayerNorm / mix kernels under the other's MFMA-bound GEMM / attention kernels) than as ONE batched B = 3 pass?
Timing experiment only: the two groups' token-side weight gradients race on the flat gradient buffer (not used)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from modaltune_amd import synth
from modaltune_amd.config import ModelConfig
from modaltune_amd.engine import Engine
L = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
dev = torch.device("cuda", 0)
cfg = ModelConfig(); sizes = synth.toy_group_sizes()
eng = Engine(cfg, sizes, dev)
eng.load_state_dict(synth.synth_state_dict(cfg, sizes, seed=0))
eng.set_stochastic(True, seed=1)
inp = synth.synth_inputs(L, sizes, seed=1, grid=128)
x = torch.from_numpy(inp["x"]).to(dev).half().reshape(L, -1).contiguous(); coords = torch.from_numpy(inp["coords"]).to(dev)
genes = [torch.from_numpy(a).to(dev) for a in inp["genes"]]
eye = torch.eye(3, device=dev)
dl = torch.randn(3, 256, device=dev) * 100.0
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def batched():
    logits = eng.forward(x, coords, genes, eye, need_grad=True, fresh=True)
    call = eng.last_call
    eng.backward(dl, call=call)


def split(offset_first=False):
    main = torch.cuda.current_stream()
    s1.wait_stream(main); s2.wait_stream(main)
    calls = {}
    with torch.cuda.stream(s1):
        eng.forward(x, coords, genes, eye[:2], need_grad=True, fresh=True); calls[1] = eng.last_call
    with torch.cuda.stream(s2):
        eng.forward(x, coords, genes, eye[2:], need_grad=True, fresh=True); calls[2] = eng.last_call
    with torch.cuda.stream(s1):
        eng.backward(dl[:2], call=calls[1])
    with torch.cuda.stream(s2):
        eng.backward(dl[2:], call=calls[2])
    main.wait_stream(s1); main.wait_stream(s2)


s3 = torch.cuda.Stream()


def split3(share_x0=True):
    main = torch.cuda.current_stream()
    sts = (s1, s2, s3)
    share = {} if share_x0 else None
    calls = []
    if share is not None:       # the task-independent patch embedding once, on the main stream
        pass
    for i, st in enumerate(sts):
        st.wait_stream(main)
        if i > 0 and share is not None:
            st.wait_stream(sts[0])          # (x0 is produced by the first group's stream)
        with torch.cuda.stream(st):
            eng.forward(x, coords, genes, eye[i:i + 1], need_grad=True, fresh=True, share=share); calls.append(eng.last_call)
    for i, st in enumerate(sts):
        with torch.cuda.stream(st):
            eng.backward(dl[i:i + 1], call=calls[i])
    for st in sts:
        main.wait_stream(st)


def two_streams_three_calls():
    """Three B = 1 engine passes on TWO streams, balanced by work: A: F0, F1, B1 | B: F2, B2, B0 (B0 needs F0: an event)."""
    main = torch.cuda.current_stream()
    s1.wait_stream(main); s2.wait_stream(main)
    calls = {}
    with torch.cuda.stream(s1):
        eng.forward(x, coords, genes, eye[0:1], need_grad=True, fresh=True); calls[0] = eng.last_call
        f0 = torch.cuda.Event(); f0.record(s1)
        eng.forward(x, coords, genes, eye[1:2], need_grad=True, fresh=True); calls[1] = eng.last_call
    with torch.cuda.stream(s2):
        eng.forward(x, coords, genes, eye[2:3], need_grad=True, fresh=True); calls[2] = eng.last_call
    with torch.cuda.stream(s1):
        eng.backward(dl[1:2], call=calls[1])
    with torch.cuda.stream(s2):
        eng.backward(dl[2:3], call=calls[2])
        s2.wait_event(f0)
        eng.backward(dl[0:1], call=calls[0])
    main.wait_stream(s1); main.wait_stream(s2)


def sequential_split():      # the same two groups one after the other on ONE stream: what the split costs without any overlap
    eng.forward(x, coords, genes, eye[:2], need_grad=True, fresh=True); c1 = eng.last_call
    eng.forward(x, coords, genes, eye[2:], need_grad=True, fresh=True); c2 = eng.last_call
    eng.backward(dl[:2], call=c1); eng.backward(dl[2:], call=c2)


def timeit(fn, n=6):
    for _ in range(2):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for rep in range(2):
    print(f"batched B=3: {timeit(batched):.2f} ms | B=2 + B=1 on one stream: {timeit(sequential_split):.2f} ms | on two streams: {timeit(split):.2f} ms"
          f" | three B=1 calls balanced over two streams: {timeit(two_streams_three_calls):.2f} ms"
          f" | three B=1 passes on three streams: {timeit(lambda: split3(False)):.2f} ms, sharing x0: {timeit(lambda: split3(True)):.2f} ms", flush=True)
