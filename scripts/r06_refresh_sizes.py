"""cloud sizes of the forwards of a guided run (NIRRT_REFRESH_LOG=1): how many size groups a refresh has, how many clouds sit in groups of one"""
import os, sys, collections
os.environ["NIRRT_REFRESH_LOG"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-ttfs", "--no-secondary", "--steps", "1", "--warmup", "0"] + sys.argv[1:]
import bench
from nirrt_star_amd import batch
logs = []
orig = batch.Guidance.__init__
def init(self, *a, **k):
    orig(self, *a, **k)
    logs.append(self)
batch.Guidance.__init__ = init
bench.main()
for g in logs:
    L = g.size_log
    groups = [len(d) for d in L]
    full = sum(d.get(2048, 0) for d in L)
    small = sum(v for d in L for n, v in d.items() if n != 2048)
    ones = sum(1 for d in L for n, v in d.items() if v == 1)
    hist = collections.Counter(n // 256 * 256 for d in L for n, v in d.items() for _ in range(v))
    print("refreshes %d, forwards %d, clouds of 2048 points %d, smaller %d, forwards over ONE cloud %d" % (len(L), sum(groups), full, small, ones), file=sys.stderr)
    print("clouds by size bucket:", sorted(hist.items()), file=sys.stderr)
    print("groups per refresh:", sorted(collections.Counter(groups).items()), file=sys.stderr)
