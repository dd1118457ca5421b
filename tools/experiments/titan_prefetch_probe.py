"""Where does the look-ahead gridding of the TITAN configuration run?  Wraps TitanEngine.stage_slide / prestage_slide with timing events:
for each step t prints (ms from the take-over of slide t on the main stream, i.e. the start of step t's graph) to (gridding of slide t + 1
done on the side stream), and the host time blocked in finish()."""
import os, sys, time, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import bench
from modaltune_amd import titan as T

marks, blocked = [], []
orig_stage, orig_pre, orig_finish = T.TitanEngine.stage_slide, T.TitanEngine.prestage_slide, T.GridStage.finish


def stage(self, x, coords, psz=1024):
    r = orig_stage(self, x, coords, psz)
    e = torch.cuda.Event(enable_timing=True); e.record()
    marks.append(["take", e])
    return r


def pre(self, x, coords, psz=1024, ready=None):
    e0 = torch.cuda.Event(enable_timing=True)
    orig_pre(self, x, coords, psz, ready)
    with torch.cuda.stream(self._pre_stream):
        e0.record()
    marks.append(["grid_done", e0])


def finish(self):
    t0 = time.perf_counter()
    r = orig_finish(self)
    blocked.append(time.perf_counter() - t0)
    return r


T.TitanEngine.stage_slide, T.TitanEngine.prestage_slide, T.GridStage.finish = stage, pre, finish
# MT_PRE_MODE (one slide, so stale gridding results stay valid): after 12 full look-aheads the side stream gets only a part of the work
#   full (default) | none: just the count copy + event | copies: the two input copies | kernels: the gridding kernels without the copies
MODE = os.environ.get("MT_PRE_MODE", "full")
orig_launch = T.GridStage.launch
ncall = [0]


def launch(self, features, coords, psz, err):
    ncall[0] += 1
    if MODE == "full" or ncall[0] <= 12 or torch.cuda.current_stream() == torch.cuda.default_stream():
        return orig_launch(self, features, coords, psz, err)
    f = features.reshape(-1, features.shape[-1]); L, C = f.shape
    if MODE == "copies":
        self.f[:L].copy_(f, non_blocking=True); self.c[:L].copy_(coords.reshape(-1, 2).to(self.dev), non_blocking=True)
    if MODE in ("kernels", "grid", "sums", "order"):
        from modaltune_amd import ops
        if MODE in ("kernels", "grid"):
            ops.titan_grid(self.c, L, float(psz), self.cells, self.dims, err)
        if MODE in ("kernels", "sums"):
            ops.titan_cell_sums(self.f, self.cells, L, C, self.first, self.nxt, self.sums, self.nz)
        if MODE in ("kernels", "order"):
            ops.titan_token_order(self.cells, self.first, self.nz, L, self.pos, self.cells_tok, self.count)
    if MODE != "nod2h":
        self.count_host.copy_(self.count, non_blocking=True)
    self.done.record()
    self.L, self.Lv = L, -1


T.GridStage.launch = launch
if MODE == "paced":          # full look-ahead, but the host waits for the running step before it enqueues the next one
    inner_stage = T.TitanEngine.stage_slide

    def paced(self, x, coords, psz=1024):
        torch.cuda.current_stream().synchronize()
        return inner_stage(self, x, coords, psz)
    T.TitanEngine.stage_slide = paced
graphs = []
orig_replay = torch.cuda.CUDAGraph.replay


def replay(self):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); orig_replay(self); e1.record()
    graphs.append((e0, e1))


torch.cuda.CUDAGraph.replay = replay
sys.argv = ["bench.py", "--config", "titan", "--patches", "4096", "--steps", "12", "--warmup", "8", "--no-cpu-baseline"]
args = bench.parse_args()
out = bench.run_titan(args)
torch.cuda.synchronize()
print("mode", MODE, "ms_per_step", round(out["ms_per_step"], 3))
pairs = []
for i in range(len(marks) - 1):
    if marks[i][0] == "take" and marks[i + 1][0] == "grid_done":
        pairs.append(round(marks[i][1].elapsed_time(marks[i + 1][1]), 3))
print("take(t) -> gridding(t+1) done, ms:", pairs[-14:])
takes = [m[1] for m in marks if m[0] == "take"]
print("take(t) -> take(t+1), ms:", [round(takes[i].elapsed_time(takes[i + 1]), 3) for i in range(len(takes) - 13, len(takes) - 1)])
print("graph replay (GPU time between events around it), ms:", [round(a.elapsed_time(b), 3) for a, b in graphs[-12:]])
print("end of graph(t) -> start of graph(t+1), ms:", [round(graphs[i][1].elapsed_time(graphs[i + 1][0]), 3) for i in range(len(graphs) - 12, len(graphs) - 1)])
print("host blocked in finish(), ms:", [round(1e3 * b, 3) for b in blocked[-14:]])
