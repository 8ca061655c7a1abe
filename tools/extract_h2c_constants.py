#!/usr/bin/env python3
"""Pulls the RFC 9380 bls12-381 hash-to-curve constants (Appendix E.2 / E.3 isogeny coefficients,
section 8.8 SWU parameters) out of the reference's source as NUMBERS ONLY and writes
tools/h2c_constants.json, the single data file the constant generator (tools/gen_consts.py) and
the oracle (oracle/h2c.py) read.  Run in the container where /root/reference exists."""
import json
import os
import re

SRC = "/root/reference/src/bls12-381.ts"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "h2c_constants.json")


def main():
    ts = open(SRC).read()
    g2 = ts[ts.index("const isogenyMapG2 = isogenyMap("):ts.index("const isogenyMapG1 = isogenyMap(")]
    g1 = ts[ts.index("const isogenyMapG1 = isogenyMap("):ts.index("let G1_SWU")]
    swu = ts[ts.index("const getG1_SWU"):ts.index("function mapToG1")]

    def rows(block):
        # split the top-level coefficient lists: xNum, xDen, yNum, yDen (marked by comments)
        parts = re.split(r"//\s*(?:xNum|xDen|yNum|yDen)", block)[1:]
        assert len(parts) == 4, len(parts)
        return parts
    out = {"G1": {}, "G2": {}}
    for name, part in zip(("xnum", "xden", "ynum", "yden"), rows(g1)):
        out["G1"][name] = [str(int(h, 16)) for h in re.findall(r"'0x([0-9a-fA-F]+)'", part)]
    for name, part in zip(("xnum", "xden", "ynum", "yden"), rows(g2)):
        hx = [str(int(h, 16)) for h in re.findall(r"'0x([0-9a-fA-F]+)'", part)]
        assert len(hx) % 2 == 0
        out["G2"][name] = [[hx[i], hx[i + 1]] for i in range(0, len(hx), 2)]
    a_b = re.findall(r"'0x([0-9a-fA-F]+)'", swu)
    out["G1"]["A"], out["G1"]["B"], out["G1"]["Z"] = str(int(a_b[0], 16)), str(int(a_b[1], 16)), "11"
    out["G2"]["A"], out["G2"]["B"], out["G2"]["Z"] = ["0", "240"], ["1012", "1012"], ["-2", "-1"]
    assert "BigInt(240)" in swu and "BigInt(1012)" in swu and "BigInt(-2)" in swu and "BigInt(11)" in swu
    json.dump(out, open(OUT, "w"), indent=1)
    print({k: {n: len(v) for n, v in d.items() if isinstance(v, list)} for k, d in out.items()})


if __name__ == "__main__":
    main()
