#!/usr/bin/env python3
"""Extracts the reference's own test fixtures for the hot path into small JSON files.

Run in the build container (where /root/reference is mounted):
    python tests/golden/make_golden.py
The GPU box has no /root/reference, so tests read only the committed JSON.
Sources (data files, not code): /root/reference/test/vectors/...
  secp256k1/privates-2.txt     k:x:y  (test/secp256k1.test.ts:59-71)
  secp256k1/points.json        bitcoinjs tiny-secp256k1 vectors (test/secp256k1.test.ts:79-131),
                               incl. valid.isPoint for the SEC1 decoder (:81-88)
  secp256k1/endomorphism.json  GLV multiplyUnsafe KATs (test/nist.test.ts:550-559)
  bls12-381/zkcrypto/converted.json  i*G for G1/G2 (test/bls12-381.test.ts:1463-1535)
  ed25519/vectors.txt          cr.yp.to sign.input (test/ed25519.test.ts:50-66)
  ed25519/zip215.json          ZIP-215 verdicts (test/ed25519.test.ts:393-418)
  ed25519/edge-cases.json      (test/ed25519.test.ts:189)
  secp256k1/ecdsa.json         RFC 6979 sign / verify vectors (test/secp256k1.test.ts:133-146, :263-270)
  secp256k1/schnorr.csv        BIP-340 test vectors (test/secp256k1.test.ts:666-684)
  wycheproof/ecdsa_test.json   the secp256k1 / SHA-256 groups, DER signatures (the same cases as the
                               un-vendored acvp-vectors file of test/secp256k1.test.ts:221-261)
"""
import json
import os

REF = "/root/reference/test/vectors"
OUT = os.path.dirname(os.path.abspath(__file__))


def dump(name, obj):
    with open(os.path.join(OUT, name), "w") as f:
        json.dump(obj, f, separators=(",", ":"))
    print(name, os.path.getsize(os.path.join(OUT, name)), "bytes")


def main():
    # secp256k1
    rows = []
    for line in open(f"{REF}/secp256k1/privates-2.txt"):
        line = line.strip()
        if line:
            k, x, y = line.split(":")
            rows.append([k, x, y])
    dump("secp256k1_privates2.json", rows)
    pts = json.load(open(f"{REF}/secp256k1/points.json"))
    dump("secp256k1_points.json", {
        "valid": {k: pts["valid"][k] for k in ("pointMultiply", "pointAdd", "pointFromScalar")},
        "invalid": {k: pts["invalid"][k] for k in ("pointMultiply",)},
    })
    dump("secp256k1_endomorphism.json", json.load(open(f"{REF}/secp256k1/endomorphism.json")))
    # isPoint: compressed encodings only (33 bytes), every rejected one plus every 4th accepted one
    isp = [v for v in pts["valid"]["isPoint"] if len(v["P"]) == 66]
    dump("secp256k1_ispoint_compressed.json",
         [v for i, v in enumerate(isp) if (not v["expected"]) or i % 4 == 0])
    # bls12-381: uncompressed i*G, first 256 of each
    conv = json.load(open(f"{REF}/bls12-381/zkcrypto/converted.json"))
    dump("bls12_381_multiples.json", {
        "G1_Uncompressed": conv["G1_Uncompressed"][:256],
        "G2_Uncompressed": conv["G2_Uncompressed"][:256],
    })
    dump("bls12_381_g1_compressed.json", conv["G1_Compressed"][:256])
    dump("bls12_381_g2_compressed.json", conv["G2_Compressed"][:256])
    # ed25519: first 160 sign.input lines (messages of 0..159 bytes)
    rows = []
    for i, line in enumerate(open(f"{REF}/ed25519/vectors.txt")):
        if i >= 160:
            break
        parts = line.strip().split(":")
        sk_pk, pk, msg, sig_msg = parts[0], parts[1], parts[2], parts[3]
        rows.append({"sk": sk_pk[:64], "pk": pk, "msg": msg, "sig": sig_msg[:128]})
    dump("ed25519_vectors.json", rows)
    # hash_to_field known answers (expand_message_xmd over SHA-256 into the bls12-381 scalar field,
    # test/bls12-381.test.ts:1281-1293) and SEC1 pointCompress vectors (test/secp256k1.test.ts:104-113)
    rows = [l.rstrip("\n").split(":") for l in open(f"{REF}/bls12-381/bls12-381-scalar-xmd-sha256-test-vectors.txt") if ":" in l]
    dump("bls12_381_scalar_xmd.json", {"DST": "QUUX-V01-CS02-with-BLS12381SCALAR_XMD:SHA-256_SSWU_RO_",
                                       "vectors": [{"msg": r[0], "expected": r[1]} for r in rows]})
    dump("secp256k1_point_compress.json", pts["valid"]["pointCompress"])
    # hash-to-curve: EIP-2537 mapToCurve vectors (test/bls12-381.test.ts:1605-1626) and the head of the
    # priv:msg:sig signature vectors that pin hashToCurve end to end (:953-966, :1003-1012)
    dump("bls12_381_eip2537.json", json.load(open(f"{REF}/bls12-381/eip2537.json")))
    sig = {}
    for g in ("g1", "g2"):
        rows = [l.strip().split(":") for l in open(f"{REF}/bls12-381/bls12-381-{g}-test-vectors.txt") if l.strip()]
        sig[g] = [{"priv": r[0], "msg": r[1], "sig": r[2]} for r in rows[:48]]
    dump("bls12_381_sig_vectors.json", sig)
    # FFT known answers are inline in test/fft.test.ts (:155-183 roots/brp tables, :221-237 Basic FFT)
    import re
    ts = open(f"{REF}/../fft.test.ts").read()

    def bigints(block):
        return [str(int(x)) for x in re.findall(r"(\d+)n", block)]
    i0 = ts.index("roots = fft.rootsOfUnity(bls12_381.fields.Fr, 7n);\n      eql(")
    i1 = ts.index("'bls12_381 roots'")
    i2 = ts.index("'bls12_381 brp'")
    j0 = ts.index("it('Basic FFT'")
    j1 = ts.index("eql(fftFr.direct(input), exp);")
    basic = ts[j0:j1]
    dump("fft_kat.json", {
        "generator": "7",
        "roots3": bigints(ts[i0:i1])[1:],            # drop the generator literal 7n
        "brp3": bigints(ts[i1:i2]),
        "basic_input": bigints(basic[basic.index("const input"):basic.index("const exp")]),
        "basic_exp": bigints(basic[basic.index("const exp"):]),
    })
    dump("ed25519_zip215.json", json.load(open(f"{REF}/ed25519/zip215.json")))
    dump("ed25519_edge_cases.json", json.load(open(f"{REF}/ed25519/edge-cases.json")))
    # Wycheproof (old format, 145 cases in 51 key groups): pk, msg, sig, expected verdict (test/ed25519.test.ts:420-444)
    wy = json.load(open(f"{REF}/ed25519/ed25519_test_OLD.json"))
    dump("ed25519_wycheproof_old.json", [{"pk": g["key"]["pk"], "msg": t["msg"], "sig": t["sig"], "result": t["result"],
                                          "comment": t["comment"]} for g in wy["testGroups"] for t in g["tests"]])
    # ECDSA: every invalid.verify case, every 5th valid (d, m, signature) vector (404 of 2019)
    ec = json.load(open(f"{REF}/secp256k1/ecdsa.json"))
    wp = json.load(open(f"{REF}/wycheproof/ecdsa_test.json"))
    groups = [{"pub": g["key"]["uncompressed"],
               "tests": [{"msg": t["msg"], "sig": t["sig"], "result": t["result"], "comment": t["comment"]} for t in g["tests"]]}
              for g in wp["testGroups"] if g["key"]["curve"] == "secp256k1" and g["sha"] == "SHA-256"]
    dump("secp256k1_ecdsa.json", {"valid": ec["valid"][::5], "invalid_verify": ec["invalid"]["verify"], "wycheproof": groups})
    # Wycheproof ECDH on secp256k1 (test/secp256k1.test.ts:272-292): valid cases whose key is the plain
    # SubjectPublicKeyInfo of an uncompressed point (the reference's derToPub takes the trailing 65 bytes)
    eh = json.load(open(f"{REF}/wycheproof/ecdh_test.json"))
    ecdh_rows = []
    for g in eh["testGroups"]:
        if g.get("curve") != "secp256k1":
            continue
        for t in g["tests"]:
            pub, priv = t["public"][-130:], t["private"]
            if t["result"] == "valid" and pub.startswith("04") and len(t["public"]) == 176:
                priv = priv[2:] if len(priv) == 66 and priv.startswith("00") else priv
                if len(priv) <= 64:
                    ecdh_rows.append({"pub": pub, "priv": priv.rjust(64, "0"), "shared": t["shared"]})
    dump("secp256k1_ecdh.json", ecdh_rows)
    # BIP-340 Schnorr vectors (test/secp256k1.test.ts:666-684): index, secret key, public key, aux, message, signature, result
    import csv
    rows = list(csv.reader(open(f"{REF}/secp256k1/schnorr.csv")))[1:]
    dump("secp256k1_schnorr.json", [{"pub": r[2], "msg": r[4], "sig": r[5], "result": r[6] == "TRUE", "comment": r[7]} for r in rows if len(r) >= 7])


if __name__ == "__main__":
    main()
