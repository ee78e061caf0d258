"""Host time per phase of the training step (no device synchronisation inside a step: these are ISSUE times), and the caching allocator's
device allocations per step (a hipMalloc inside a step stalls the host for ~1 ms)."""
import os, sys, time, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import __graft_entry__ as g
g.build()
from diffspectra_amd import losses as Lh, train_engine as TE, spec_train as ST

acc = collections.defaultdict(float)
cnt = collections.Counter()

def wrap(cls, name, label=None):
    f = getattr(cls, name)
    lab = label or f"{cls.__name__}.{name}"
    def w(*a, **k):
        t = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            acc[lab] += time.perf_counter() - t
            cnt[lab] += 1
    setattr(cls, name, w)

wrap(TE.DmtTrainGraph, "forward"); wrap(TE.DmtTrainGraph, "backward"); wrap(TE.DmtTrainGraph, "loss"); wrap(TE.DmtTrainGraph, "prepare_weights")
wrap(TE.DmtTrainGraph, "scatter_cat_grads"); wrap(ST.SpecTrainGraph, "forward"); wrap(ST.SpecTrainGraph, "backward")
wrap(Lh.HipTrainer, "graphs"); wrap(Lh.HipTrainer, "layout"); wrap(Lh.HipTrainer, "stage"); wrap(Lh.FusedAdamW, "step"); wrap(Lh.FusedAdamW, "zero_grad")
wrap(TE.Ops, "join_dw")
orig_bw = torch.Tensor.backward
def bw(self, *a, **k):
    t = time.perf_counter()
    try:
        return orig_bw(self, *a, **k)
    finally:
        acc["Tensor.backward"] += time.perf_counter() - t; cnt["Tensor.backward"] += 1
torch.Tensor.backward = bw
orig_get = Lh.get_step_fn
def get_step_fn(*a, **k):
    sf = orig_get(*a, **k)
    seen = [0]
    def timed(state, batch):
        seen[0] += 1
        if seen[0] == 9:                       # steady state only: drop the warm-up calls (caches, first allocations)
            acc.clear(); cnt.clear()
        if seen[0] > 25:                       # the instrumented roofline steps after the timed region
            return sf(state, batch)
        t = time.perf_counter()
        n0 = torch.cuda.memory_stats().get("num_device_alloc", 0)
        try:
            return sf(state, batch)
        finally:
            acc["step_fn"] += time.perf_counter() - t; cnt["step_fn"] += 1
            acc["device_allocs"] += torch.cuda.memory_stats().get("num_device_alloc", 0) - n0
    return timed
Lh.get_step_fn = get_step_fn
args = bench.parse_args(["--mode", "train", "--steps", "20", "--warmup", "5", "--no-cpu-baseline"])
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
orig_measure_sync = torch.cuda.synchronize
line = bench.train_measure(args, 1, 0, dev, 20, 5, False)
print("ms_per_step", line["ms_per_step"])
n = cnt["step_fn"]
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"{k:36s} {1e3 * v / n:8.3f} ms per step  ({cnt[k] / n:.2f} calls per step)")
