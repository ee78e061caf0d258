"""Host-side cost of one training step: cProfile over a few steps of bench.py's train loop (is the step bound by issuing launches?)."""
import cProfile, pstats, io, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
args = bench.parse_args(["--mode", "train", "--steps", "6", "--warmup", "3", "--no-cpu-baseline"])
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
import __graft_entry__ as g
g.build()
pr = cProfile.Profile()
orig = bench.train_measure

def wrapped(*a, **k):
    pr.enable()
    try:
        return orig(*a, **k)
    finally:
        pr.disable()

line = wrapped(args, 1, 0, dev, 6, 3, False)
print("ms_per_step", line["ms_per_step"], file=sys.stderr)
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
