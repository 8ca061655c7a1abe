import os, sys, time, json
ROOT = "/root/repo" if os.path.isdir("/root/repo/tools") else os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch, bench
from noble_curves_amd import get_engine
from noble_curves_amd._native import BLS12_381_G1
from oracle.curves import BlsG1
dev = torch.device("cuda", 0); st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st); s = st.cuda_stream
eng = get_engine(0)
cid, O = BLS12_381_G1, BlsG1
n = 1 << 20
pts, ks = bench.gen_points(eng, cid, O, n, 0x1234567 + 7, 0x6789, dev, s)
sc = bench.gen_scalars(n, 254, 5, dev); sc[::17] = 0
rs = eng.upload_points(cid, pts.cpu().numpy())
torch.cuda.synchronize()
G = 8
def pipelined(submit, collect, depth, jobs):
    for warm in (True, False):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(jobs):
            lane = i % depth
            if i >= depth: collect(lane)
            submit(lane, i)
        for i in range(jobs, jobs + depth): collect(i % depth)
        dt = (time.perf_counter() - t0) * 1e3
    return dt / jobs
for rep in range(2):
    for name, sarg in (("stream=None", None), ("stream=bench stream", s)):
        for depth in (3,):
            ms = pipelined(lambda lane, i: eng.msm_async_submit(lane, cid, n, 0, sc.data_ptr(), sarg, rs, eng.async_part(i % G, G)),
                           lambda lane: eng.msm_async_collect_slot(lane, cid), depth, 40)
            print(name, "depth", depth, round(ms, 4), flush=True)
    # the bench's own harness
    coll = {}
    def sub(lane, i):
        eng.msm_async_submit(lane, cid, n, 0, sc.data_ptr(), s, rs, eng.async_part(i % G, G)); coll[lane] = i % G
    def col(lane):
        return coll[lane], eng.msm_async_collect_slot(lane, cid)
    seen = {}
    pw, _ = bench.time_pipelined(sub, col, 3, 40, 5, False, lambda r: seen.__setitem__(r[0], r[1]))
    print("bench.time_pipelined", round(pw / 40 * 1e3, 4), flush=True)

# ---- does the lanes' earlier use by whole generic MSMs (bench.py runs those first) change the share's in-flight time?
def chk(r): pass
for depth in (2, 3):
    pw, _ = bench.time_pipelined(lambda lane, i: eng.msm_async_submit(lane, cid, n, pts.data_ptr(), sc.data_ptr(), s),
                                 lambda lane: eng.msm_async_collect(lane, cid), depth, 20, 5, False, chk)
    print("generic pipelined depth", depth, round(pw / 20 * 1e3, 4), flush=True)
seen = {}
pw, _ = bench.time_pipelined(sub, col, 3, 40, 5, False, lambda r: seen.__setitem__(r[0], r[1]))
print("share after generic lanes: bench.time_pipelined", round(pw / 40 * 1e3, 4), flush=True)
# the sync share calls in between, as bench.py does
slots = [eng.msm_shard_windows_local_dev(cid, n, r, G, 0, sc.data_ptr(), s, rs) for r in range(G)]
st_l = bench.time_steps(lambda: eng.msm_shard_windows_local_dev(cid, n, 0, G, 0, sc.data_ptr(), s, rs), 20, 5, False)
print("local part0 sync", round(st_l[0] / 20 * 1e3, 4))
pw, _ = bench.time_pipelined(sub, col, 3, 40, 5, False, lambda r: seen.__setitem__(r[0], r[1]))
print("share after sync share calls: bench.time_pipelined", round(pw / 40 * 1e3, 4), flush=True)
