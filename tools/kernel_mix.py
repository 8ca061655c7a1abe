#!/usr/bin/env python3
"""Static picture of the kernels in a HIP library: code size, span of the longest loop (backward branch), registers /
spills / scratch, and the instruction mix of the kernel text (v_mad_u64_u32, other VALU, LDS, memory, scalar, s_nop) from
the disassembly of the code objects.  Counts are STATIC (every instruction once, loops not weighted).
    python tools/kernel_mix.py noble-curves_amd/libncg.so [substring ...] > profiles/rNN_kernel_mix.md"""
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
import kernel_resources as kr  # noqa: E402

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def classify(op):
    if op == "v_mad_u64_u32":
        return "mad"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "scratch_", "buffer_", "flat_")):
        return "mem"
    if op == "s_nop":
        return "nop"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    blob = open(sys.argv[1], "rb").read()
    pats = sys.argv[2:]
    rows = []
    for co in kr.bundles(blob):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            notes = subprocess.run([kr.READELF, "--notes", f.name], capture_output=True, text=True).stdout
            dis = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True).stdout
        meta = {}
        for blk in notes.split("- .agpr_count:")[1:]:
            nm = re.search(r"\.name:\s+(\S+)", blk)
            if nm:
                g = lambda k: int((re.search(r"\.%s:\s+(\d+)" % k, blk) or [0, 0])[1])  # noqa: E731
                meta[nm.group(1)] = (g("vgpr_count"), g("vgpr_spill_count"), g("private_segment_fixed_size"))
        cur, stats = None, {}
        for line in dis.splitlines():
            m = re.match(r"^([0-9a-f]+) <(.+)>:", line)
            if m:
                cur = m.group(2)
                stats[cur] = {"start": int(m.group(1), 16), "end": int(m.group(1), 16), "back": 0, "mix": {}}
                continue
            if cur is None:
                continue
            m = re.match(r"\s+(\S+)(.*?)//\s+([0-9A-Fa-f]+):", line)
            if not m:
                continue
            op, rest, addr = m.group(1), m.group(2), int(m.group(3), 16)
            st = stats[cur]
            st["end"] = addr
            c = classify(op)
            st["mix"][c] = st["mix"].get(c, 0) + 1
            if op.startswith(("s_cbranch", "s_branch")):
                off = re.search(r"(\d+)", rest)
                if off and int(off.group(1)) > 32767:
                    st["back"] = max(st["back"], (65536 - int(off.group(1))) * 4)
        for k, st in stats.items():
            if k not in meta:
                continue
            dem = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip().split("(")[0]
            dem = dem.replace("void ncg::", "").replace("ncg::", "")
            if pats and not any(p in dem for p in pats):
                continue
            rows.append((dem, st, meta[k]))
    rows.sort(key=lambda r: r[0])
    print("| kernel | code KB | longest loop KB | VGPR | spills | scratch B | v_mad_u64_u32 | other VALU | LDS | memory | scalar | s_nop |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    seen = set()
    for dem, st, (vg, sp, sc) in rows:
        key = (dem, st["end"] - st["start"])
        if key in seen:
            continue
        seen.add(key)
        mx = st["mix"]
        print("| `%s` | %.1f | %.1f | %d | %d | %d | %d | %d | %d | %d | %d | %d |" % (
            dem, (st["end"] - st["start"]) / 1024, st["back"] / 1024, vg, sp, sc, mx.get("mad", 0), mx.get("valu", 0),
            mx.get("lds", 0), mx.get("mem", 0), mx.get("salu", 0), mx.get("nop", 0)))


if __name__ == "__main__":
    main()
