import torch, sys
M=30003
junk = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
g = torch.Generator(device="cuda").manual_seed(0)
for N,K in [(3072,768),(768,3072),(2304,768),(768,768),(768,2304)]:
    A = (torch.randn(M, K, device="cuda", generator=g) * 0.5).half(); W = (torch.randn(N, K, device="cuda", generator=g) * 0.05).half()
    C = torch.zeros(M, N, device="cuda", dtype=torch.float16)
    r=[]
    for cold in (True, False):
        ts=[]
        for it in range(8):
            if cold: junk.fill_(float(it))
            e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
            e0.record(); torch.matmul(A, W.t(), out=C); e1.record(); torch.cuda.synchronize()
            if it>=2: ts.append(e0.elapsed_time(e1)*1e3)
        r.append(sum(ts)/len(ts))
    print(f"hipBLASLt (torch.matmul) N={N} K={K}: cold {r[0]:.0f} us warm {r[1]:.0f} us", flush=True)
