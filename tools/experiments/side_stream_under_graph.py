"""Does work on a side stream, enqueued right AFTER a hipGraph launch, run under that graph -- or behind it?
A graph of N dependent kernels (each ~40 us) is launched; then a side stream gets one small kernel and an event.  Prints when the side
event completed relative to the graph's start / end (HIP event times), for a side stream of default and of high priority, and for the
side work enqueued BEFORE the graph launch."""
import sys, torch
dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
a = torch.randn(2048, 2048, device=dev, dtype=torch.float16)
b = torch.randn(2048, 2048, device=dev, dtype=torch.float16)
c = torch.empty_like(a)
small = torch.zeros(1 << 16, device=dev)
main = torch.cuda.current_stream() if len(sys.argv) > 2 and sys.argv[2] == "default" else torch.cuda.Stream()
cap = torch.cuda.Stream()


def trial(side, order):
    with torch.cuda.stream(main):
        g = torch.cuda.CUDAGraph()
        torch.mm(a, b, out=c); torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=cap if main == torch.cuda.default_stream() else main):
            for _ in range(N):
                torch.mm(a, b, out=c)
        g.replay(); torch.cuda.synchronize()
        res = []
        for _ in range(3):
            e0, e1, es = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record(main)
            side.wait_event(e0)
            if order == "before":
                with torch.cuda.stream(side):
                    small.add_(1.0); es.record(side)
            g.replay()
            e1.record(main)
            if order == "after":
                with torch.cuda.stream(side):
                    small.add_(1.0); es.record(side)
            torch.cuda.synchronize()
            res.append((round(e0.elapsed_time(es), 3), round(e0.elapsed_time(e1), 3)))
    return res


for name, side in (("default priority", torch.cuda.Stream()), ("high priority", torch.cuda.Stream(priority=-1))):
    for order in ("after", "before"):
        print(f"side stream {name}, side work enqueued {order} the graph launch: (side done ms, graph done ms) =", trial(side, order))


def chained(side):
    """graph A; event e on main; graph B; side waits e, runs a small kernel: when does it run relative to B's start / end?"""
    with torch.cuda.stream(main):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=cap if main == torch.cuda.default_stream() else main):
            for _ in range(N):
                torch.mm(a, b, out=c)
        g.replay(); torch.cuda.synchronize()
        res = []
        for _ in range(3):
            e0, ea, eb, es = (torch.cuda.Event(enable_timing=True) for _ in range(4))
            e0.record(main)
            g.replay()                 # A
            ea.record(main)            # "the buffers are free": behind A
            g.replay()                 # B
            eb.record(main)
            side.wait_event(ea)
            with torch.cuda.stream(side):
                small.add_(1.0); es.record(side)
            torch.cuda.synchronize()
            res.append(tuple(round(e0.elapsed_time(x), 3) for x in (ea, es, eb)))
    return res


print("chained (A done, side done, B done) ms:", chained(torch.cuda.Stream()))


def back_to_back(nsmall=0):
    """Duration of graph replay A with nothing enqueued behind it vs with a second replay (and nsmall small eager launches) enqueued
    behind it while it runs."""
    with torch.cuda.stream(main):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=cap if main == torch.cuda.default_stream() else main):
            for _ in range(N):
                torch.mm(a, b, out=c)
        g.replay(); torch.cuda.synchronize()
        alone, followed = [], []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(main); g.replay(); e1.record(main)
            torch.cuda.synchronize()
            alone.append(round(e0.elapsed_time(e1), 3))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(main); g.replay(); e1.record(main)
            for _ in range(nsmall):
                small.add_(1.0)
            g.replay()
            torch.cuda.synchronize()
            followed.append(round(e0.elapsed_time(e1), 3))
    return alone, followed


print("graph alone / with another replay enqueued behind it, ms:", back_to_back())
print("same, + 10 small eager launches in between:", back_to_back(10))
