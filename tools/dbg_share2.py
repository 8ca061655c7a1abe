import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
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
coll = {}
def sub(lane, i):
    eng.msm_async_submit(lane, cid, n, 0, sc.data_ptr(), s, rs, eng.async_part(i % G, G)); coll[lane] = i % G
def col(lane):
    return coll[lane], eng.msm_async_collect_slot(lane, cid)
seen = {}
def share(tag):
    pw, _ = bench.time_pipelined(sub, col, 3, 40, 5, False, lambda r: seen.__setitem__(r[0], r[1]))
    pw, _ = bench.time_pipelined(sub, col, 3, 40, 5, False, lambda r: seen.__setitem__(r[0], r[1]))
    print("[dbg2] %-44s share ms/job %.4f" % (tag, pw / 40 * 1e3), flush=True)

def submit_cost(tag):
    ts = []
    for i in range(60):
        t0 = time.perf_counter(); sub(0, i); t1 = time.perf_counter(); col(0); t2 = time.perf_counter()
        ts.append(((t1 - t0) * 1e6, (t2 - t1) * 1e6))
    ts = ts[10:]
    a = sorted(x for x, _ in ts); b = sorted(y for _, y in ts)
    print("[dbg] %s: submit call median %.1f us, collect median %.1f us" % (tag, a[len(a) // 2], b[len(b) // 2]), flush=True)

import sys
submit_cost('fresh process')
print('[dbg2] os threads', len(os.listdir('/proc/self/task')))
share("fresh process")
for _ in range(30): eng.msm_dev(cid, n, pts.data_ptr(), sc.data_ptr(), s)
share("after 30 synchronous MSMs (finish helpers up)")
os.environ["NCG_NO_FINISH_THREADS"] = "1"
pts_h, sc_h = pts.cpu().numpy(), sc.cpu().numpy()
eng.host_register(pts_h); eng.host_register(sc_h)
for _ in range(5): eng.msm(cid, pts_h, sc_h)
eng.host_unregister(pts_h); eng.host_unregister(sc_h)
share("after the host-pointer MSM on pinned buffers")
for G2 in (2, 4):
    def sub2(lane, i, G2=G2):
        eng.msm_async_submit(lane, cid, n, 0, sc.data_ptr(), s, rs, eng.async_part(i % G2, G2)); coll[lane] = i % G2
    pw, _ = bench.time_pipelined(sub2, col, 3, 40, 5, False, lambda r: None)
    print("[dbg2] G=%d share ms/job %.4f" % (G2, pw / 40 * 1e3), flush=True)
share("after the G = 2 and G = 4 pipelines on the same lanes")
for G2 in (2, 4, 8):
    slots = [eng.msm_shard_windows_local_dev(cid, n, r, G2, 0, sc.data_ptr(), s, rs) for r in range(G2)]
    stack = np.stack(slots)
    for _ in range(10): eng.msm_shard_combine(cid, n, stack, s)
share("after the synchronous share + combine calls")
