#!/usr/bin/env python3
"""Generates noble-curves_amd/csrc/consts_gen.hpp: Montgomery constants (32-bit limbs,
little-endian) for the three base fields and the curve constants of the hot path.
Values restate reference src/secp256k1.ts:48-64, src/ed25519.ts:49-65,102-104,
src/bls12-381.ts:134-158,321-345.  Run: python tools/gen_consts.py
"""
import os

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "noble-curves_amd", "csrc", "consts_gen.hpp")


def limbs(x, n):
    return [(x >> (32 * i)) & 0xFFFFFFFF for i in range(n)]


def arr(name, x, n):
    return "  static constexpr uint32_t %s[%d] = {%s};\n" % (
        name, n, ", ".join("0x%08xu" % l for l in limbs(x, n)))


def field(name, p, n, extra=None, fold=None):
    """fold = (c_lo, c_hi, final_subs): plain residues, 2^256 folded as c (special-form prime);
    otherwise Montgomery constants with R = 2^(32 n)."""
    R = 1 if fold else 1 << (32 * n)
    inv = (-pow(p, -1, 1 << 32)) % (1 << 32)
    s = "struct %s {\n  static constexpr int N = %d;\n" % (name, n)
    s += arr("P", p, n)
    s += "  static constexpr uint32_t INV = 0x%08xu;  // -p^-1 mod 2^32\n" % inv
    s += arr("R1", R % p, n)        # Montgomery form of 1
    s += arr("R2", R * R % p, n)    # to-Montgomery multiplier
    s += "  static constexpr bool TOP_SPARE = %s;  // top bit of p clear: CIOS carries never overflow\n" % (
        "true" if p < (1 << (32 * n - 1)) else "false")
    if fold:
        s += "  static constexpr bool FOLD = true;   // plain residues; R1 = R2 = 1\n"
        s += "  static constexpr uint32_t FOLD_LO = 0x%08xu;\n  static constexpr uint32_t FOLD_HI = %du;\n" % (fold[0], fold[1])
        s += "  static constexpr int FINAL_SUBS = %d;\n" % fold[2]
    else:
        s += "  static constexpr bool FOLD = false;  // Montgomery form\n"
    for k, v in (extra or {}).items():
        s += arr(k, v * R % p, n)   # constants stored in Montgomery form
    s += "};\n\n"
    return s


def limbs29(x, n):
    return [(x >> (29 * i)) & ((1 << 29) - 1) for i in range(n)]


def arr29(name, x, n):
    return "  static constexpr uint32_t %s[%d] = {%s};\n" % (
        name, n, ", ".join("0x%08xu" % l for l in limbs29(x, n)))


def field29(name, p, n, extra=None):
    """Montgomery constants for radix 2^29 with n limbs (R = 2^(29 n))."""
    R = 1 << (29 * n)
    inv = (-pow(p, -1, 1 << 29)) % (1 << 29)
    s = "struct %s {\n  static constexpr int N = %d;\n  static constexpr int BITS = 29;\n" % (name, n)
    s += arr29("P", p, n)
    s += "  static constexpr uint32_t INV = 0x%08xu;  // -p^-1 mod 2^29\n" % inv
    s += arr29("R1", R % p, n)
    s += arr29("R2", R * R % p, n)
    # 2^k * p for k = 0..12 (subtraction offsets), normalised limbs
    s += "  static constexpr uint32_t PMUL[13][%d] = {\n" % n
    for k in range(13):
        s += "    {%s},\n" % ", ".join("0x%08xu" % l for l in limbs29(p << k, n))
    s += "  };\n"
    for k, v in (extra or {}).items():
        s += arr29(k, v * R % p, n)
    if n % 2 == 0:  # host-side twin in radix 2^58 (same R): limbs pair up
        inv58 = (-pow(p, -1, 1 << 58)) % (1 << 58)
        s += "  static constexpr uint64_t P58[%d] = {%s};\n" % (
            n // 2, ", ".join("0x%016xull" % ((p >> (58 * i)) & ((1 << 58) - 1)) for i in range(n // 2)))
        s += "  static constexpr uint64_t INV58 = 0x%016xull;  // -p^-1 mod 2^58\n" % inv58
    s += "};\n\n"
    return s


def field9(name, p, extra=None):
    """Constants of the carry-free radix-2^29 form of a 256-bit special prime (fe9.hpp): 9 limbs, plain
    residues, 2^261 folded as C = C0 + C1*2^29; BIAS[K] = a multiple of p whose limbs are all >= K*U
    (U = 2^29 + 2^19, the limb unit of the bound types) for subtraction without borrows."""
    M = (1 << 29) - 1
    U = (1 << 29) + (1 << 19)
    C = (1 << 261) % p
    c0, c1 = C & M, C >> 29
    assert c0 < (1 << 15) and c1 < (1 << 10)
    s = "struct %s {\n" % name
    s += arr29("P", p, 9)
    s += "  static constexpr uint32_t C0 = %du, C1 = %du;  // 2^261 mod p = C0 + C1 * 2^29\n" % (c0, c1)
    s += "  static constexpr uint32_t PINV = 0x%08xu;  // p^-1 mod 2^29\n" % pow(p, -1, 1 << 29)
    s += "  static constexpr int JMAX = %d;  // a value with limbs < B*U is below JMAX*B*p\n" % (
        (U * sum(1 << (29 * i) for i in range(9)) + p - 1) // p)
    s += "  static constexpr uint32_t BIAS[8][9] = {\n"
    for k in range(8):
        base = sum((k * U) << (29 * i) for i in range(9))
        t = (-base) % p
        lim = [k * U + ((t >> (29 * i)) & M) for i in range(9)]
        assert sum(l << (29 * i) for i, l in enumerate(lim)) % p == 0 and all(k * U <= l < (k + 1) * U for l in lim)
        s += "    {%s},\n" % ", ".join("0x%08xu" % l for l in lim)
    s += "  };\n"
    for k, v in (extra or {}).items():
        s += arr29(k, v % p, 9)
    s += "};\n\n"
    return s


def main():
    kp = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEFFFFFC2F
    kn = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
    beta = 0x7AE96A2B657C07106E64479EAC3434E99CF0497512F58995C1396C28719501EE
    ep = (1 << 255) - 19
    ed = 0x52036CEE2B6FFE738CC740797779E89800700A4D4141D8AB75EB4DCA135978A3
    sqrtm1 = 19681161376707505956807079304988542015446066515923890162744021073123829784752
    bp = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
    out = "// GENERATED by tools/gen_consts.py - do not edit.\n#pragma once\n#include <stdint.h>\nnamespace ncg {\n\n"
    out += field("ParamsSecpP", kp, 8, {"BETA": beta, "B3": 21}, fold=(977, 1, 1))
    out += field("ParamsEdP", ep, 8, {"D": ed, "D2": 2 * ed % ep, "SQRT_M1": sqrtm1}, fold=(38, 0, 2))
    out += field("ParamsBlsP", bp, 12, {})
    out += field9("Fe9SecpPR", kp, {"BETA": beta})
    out += field9("Fe9EdPR", ep, {"D": ed, "D2": 2 * ed % ep, "SQRT_M1": sqrtm1})
    g1_beta = 0x5F19672FDF76CE51BA69C6076A0F77EADDB3A93BE6F89688DE17D813620A00022E01FFFFFFFEFFFE
    # G2 psi endomorphism coefficients (src/abstract/tower.ts:240-241 with base 1/(u+1),
    # src/bls12-381.ts:283): PSI_X = base^((p-1)/3), PSI_Y = base^((p-1)/2) in Fp2 = Fp[u]/(u^2+1)
    def f2mul(a, b):
        return ((a[0] * b[0] - a[1] * b[1]) % bp, (a[0] * b[1] + a[1] * b[0]) % bp)

    def f2pow(a, e):
        r = (1, 0)
        while e:
            if e & 1:
                r = f2mul(r, a)
            a = f2mul(a, a)
            e >>= 1
        return r
    ninv = pow(2, -1, bp)
    base = (ninv, (-ninv) % bp)                      # 1/(1+u) = (1-u)/2
    assert f2mul(base, (1, 1)) == (1, 0)
    psi_x = f2pow(base, (bp - 1) // 3)
    psi_y = f2pow(base, (bp - 1) // 2)
    out += field29("ParamsBls29", bp, 14, {"FOUR": 4, "G1_BETA": g1_beta, "HALF": ninv,
                                           "PSI_X_C0": psi_x[0], "PSI_X_C1": psi_x[1],
                                           "PSI_Y_C0": psi_y[0], "PSI_Y_C1": psi_y[1]})
    # hash-to-curve constants (RFC 9380 8.8, E.2, E.3; src/bls12-381.ts:668-851) in radix-2^29
    # Montgomery form, read from tools/h2c_constants.json (numbers extracted from the reference)
    import json
    hk = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "h2c_constants.json")))
    R29 = 1 << (29 * 14)

    def m29(v):
        return "{%s}" % ", ".join("0x%08xu" % l for l in limbs29(int(v) % bp * R29 % bp, 14))
    out += "struct BlsH2c {\n"
    for k in ("xnum", "xden", "ynum", "yden"):
        rows = hk["G1"][k]
        out += "  static constexpr int ISO1_%s_N = %d;\n" % (k.upper(), len(rows))
        out += "  static constexpr uint32_t ISO1_%s[%d][14] = {\n%s\n  };\n" % (
            k.upper(), len(rows), ",\n".join("    " + m29(v) for v in rows))
    for k in ("xnum", "xden", "ynum", "yden"):
        rows = hk["G2"][k]
        out += "  static constexpr int ISO2_%s_N = %d;\n" % (k.upper(), len(rows))
        out += "  static constexpr uint32_t ISO2_%s[%d][2][14] = {\n%s\n  };\n" % (
            k.upper(), len(rows), ",\n".join("    {%s, %s}" % (m29(a), m29(b)) for a, b in rows))
    z1 = int(hk["G1"]["Z"])
    c2 = pow((-z1) % bp, (bp + 1) // 4, bp)             # sqrt(-Z), sqrt_ratio_3mod4 constant c2
    assert c2 * c2 % bp == (-z1) % bp
    for name, v in (("SWU1_A", hk["G1"]["A"]), ("SWU1_B", hk["G1"]["B"]), ("SWU1_Z", z1), ("SWU1_C2", c2)):
        out += "  static constexpr uint32_t %s[14] = %s;\n" % (name, m29(v))
    for name, v in (("SWU2_A", hk["G2"]["A"]), ("SWU2_B", hk["G2"]["B"]), ("SWU2_Z", hk["G2"]["Z"])):
        out += "  static constexpr uint32_t %s[2][14] = {%s, %s};\n" % (name, m29(v[0]), m29(v[1]))
    # K = sqrt(-norm(Z)) for the G2 SWU constant Z = -(2 + u): norm(Z) = 5 is a non-residue (Z is a non-square
    # of Fp2) and so is -1, hence -5 is a residue.  Used to turn sqrt(-norm(w)) into sqrt(norm(Z w)).
    z2 = [int(v) % bp for v in hk["G2"]["Z"]]
    nz = (z2[0] * z2[0] + z2[1] * z2[1]) % bp
    kk = pow((-nz) % bp, (bp + 1) // 4, bp)
    assert kk * kk % bp == (-nz) % bp
    out += "  static constexpr uint32_t SWU2_K[14] = %s;  // sqrt(-norm(SWU2_Z)) in Fp\n" % m29(kk)
    psi2_x = f2pow(base, (bp * bp - 1) // 3)            # src/abstract/tower.ts:249
    assert psi2_x[1] == 0 and f2pow(base, (bp * bp - 1) // 2) == (bp - 1, 0)
    out += "  static constexpr uint32_t PSI2_X[14] = %s;  // in Fp\n" % m29(psi2_x[0])
    out += "};\n\n"
    out += "// exponents / thresholds for bls12-381 Fp square roots and the compressed-point sort bit\nstruct BlsFpConsts {\n"
    out += arr("SQRT_EXP", (bp + 1) // 4, 12)          # p = 3 mod 4: sqrt(a) = a^((p+1)/4)
    out += arr("SQRT_EXP_M1", (bp - 3) // 4, 12)       # a^((p-3)/4): sqrt candidate and its inverse at once
    out += arr("HALF_P", (bp - 1) // 2, 12)            # y > (p-1)/2  <=>  sort bit
    out += arr("P32", bp, 12)
    out += "};\n\n"
    # GLV constants (libsecp256k1-style 384-bit reciprocals of the reference's basis,
    # src/secp256k1.ts:58-64): g1 = round(2^384*b2/n), g2 = round(2^384*(-b1)/n)
    a1 = 0x3086D221A7D46BCDE86C90E49284EB15
    mb1 = 0xE4437ED6010E88286F547FA90ABFE4C3
    a2 = 0x114CA50F7A8E2F3F657C1108D9D44CFD8
    b2 = a1
    g1 = ((b2 << 384) + kn // 2) // kn
    g2 = ((mb1 << 384) + kn // 2) // kn
    out += "struct SecpGlv {\n"
    out += arr("G1", g1, 8) + arr("G2", g2, 8)
    out += arr("A1", a1, 5) + arr("MB1", mb1, 5) + arr("A2", a2, 5) + arr("B2", b2, 5)
    out += arr("N", kn, 8)
    out += "};\n\n"
    bls_r = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
    ed_l = 0x1000000000000000000000000000000014DEF9DEA2F79CD65812631A5CF5D3ED
    # bls12-381 scalar field Fr as a Montgomery field for the NTT (src/bls12-381.ts bls12_381_Fr;
    # src/abstract/fft.ts).  INV2 = 1/2 for the 1/N scale of the inverse transform.
    out += field("ParamsBlsR", bls_r, 8, {"INV2": (bls_r + 1) // 2})
    # Fr in radix 2^29 for the NTT butterflies (fr29.hpp): R = 2^261, r = 1 (mod 2^29) so -r^-1 = -1 and
    # the Montgomery quotient digit is the negated low limb; C255 = 2^255 - r folds the bits at and above
    # 2^255; BIAS = 3 r with limbs 0..7 in [2^29, 2^30) for the subtraction a + BIAS - t; K261 = 2^261 mod r
    # (plain 8 x 32-bit integer) moves the twiddle table from R = 2^256 to R = 2^261.
    assert bls_r % (1 << 29) == 1
    m29 = (1 << 29) - 1
    low = sum(1 << (29 * i + 29) for i in range(8))
    rest = 3 * bls_r - low
    assert rest > 0
    bias = [((rest >> (29 * i)) & m29) + (1 << 29) for i in range(8)] + [rest >> 232]
    assert sum(b << (29 * i) for i, b in enumerate(bias)) == 3 * bls_r
    out += "struct Fr29PR {\n"
    out += arr29("P", bls_r, 9) + arr29("C255", (1 << 255) - bls_r, 9)
    out += "  static constexpr uint32_t BIAS[9] = {%s};  // 3 r\n" % ", ".join("0x%08xu" % b for b in bias)
    out += arr("K261", (1 << 261) % bls_r, 8)
    out += arr29("ONE", (1 << 261) % bls_r, 9)
    out += "};\n\n"
    out += "// group orders (Fn.ORDER of the reference's curves), 8 LE limbs\nstruct Orders {\n"
    out += arr("SECP_N", kn, 8) + arr("ED_L", ed_l, 8) + arr("BLS_R", bls_r, 8)
    out += "};\n\n"
    # generators in wire format (x || y, LE 32-bit limbs; Fp2 = c0, c1)
    kgx = 0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798
    kgy = 0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8
    g1x = 0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB
    g1y = 0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1
    g2 = [0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8,
          0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E,
          0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801,
          0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE]
    def wire(vals, n):
        w = []
        for v in vals:
            w += limbs(v, n)
        return ", ".join("0x%08xu" % l for l in w)
    out += "// generators (Point.BASE) in the wire format of include/ncg.h\nstruct BasePoints {\n"
    out += "  static constexpr uint32_t SECP[16] = {%s};\n" % wire([kgx, kgy], 8)
    out += "  static constexpr uint32_t G1[24] = {%s};\n" % wire([g1x, g1y], 12)
    out += "  static constexpr uint32_t G2[48] = {%s};\n" % wire(g2, 12)
    out += "};\n\n"
    out += "}  // namespace ncg\n"
    with open(OUT, "w") as f:
        f.write(out)
    print("wrote", os.path.normpath(OUT))


if __name__ == "__main__":
    main()
