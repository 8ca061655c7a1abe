#!/usr/bin/env python3
"""Which share of the instructions that SQ_INSTS_VALU_INT64 counts is v_mad_u64_u32, per kernel of the library (static, from the
disassembly): the counter also takes v_mad_i64_i32 and v_lshl_add_u64 (profiles/r06_valu_calib.json: v_lshrrev_b64, v_mov_b64 and
the carry instructions are NOT in it).  bench.py multiplies the counter by this share to get executed multiply-adds.
    python tools/int64_share.py noble-curves_amd/libncg.so > profiles/r06_int64_mad_share.json"""
import collections, json, re, subprocess, sys, tempfile, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_resources as kr
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
INT64 = ("v_mad_u64_u32", "v_mad_i64_i32", "v_lshl_add_u64")
out = {}
for co in kr.bundles(open(sys.argv[1], "rb").read()):
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(co); f.flush()
        dis = subprocess.run([OBJDUMP, "-d", "-C", "--no-show-raw-insn", f.name], capture_output=True, text=True).stdout
    cur, st = None, collections.defaultdict(collections.Counter)
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            cur = m.group(1)
            continue
        m = re.match(r"\s+(v_\S+)", line)
        if m and cur:
            st[cur][m.group(1).replace("_e64", "").replace("_e32", "")] += 1
    for k, c in st.items():
        tot = sum(c[o] for o in INT64)
        if tot >= 32 and k.startswith("void ncg::k_"):
            name = k[5:].split("(")[0]
            out[name] = round(c["v_mad_u64_u32"] / tot, 5)
json.dump({"what": "tools/int64_share.py: v_mad_u64_u32 / (v_mad_u64_u32 + v_mad_i64_i32 + v_lshl_add_u64), static, per kernel", "kernels": out}, sys.stdout, indent=0, sort_keys=True)
