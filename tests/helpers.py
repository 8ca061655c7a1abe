"""Shared helpers for parity tests: wire-format marshalling between oracle points and the
C-ABI byte layout (include/ncg.h)."""
import json
import os

import numpy as np

from noble_curves_amd._native import (BLS12_381_G1, BLS12_381_G2, ED25519, FIELD_BYTES, POINT_BYTES, SECP256K1,
                                      ints_to_le, le_to_ints)
from oracle.curves import BlsG1, BlsG2, Ed25519, Secp256k1

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
ORACLE_CURVE = {SECP256K1: Secp256k1, BLS12_381_G1: BlsG1, BLS12_381_G2: BlsG2, ED25519: Ed25519}


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def affine_to_wire(curve, aff):
    """oracle affine tuple -> bytes (x||y little-endian; Fp2 as c0||c1)."""
    fb = FIELD_BYTES[curve]
    x, y = aff
    if curve == BLS12_381_G2:
        parts = [x[0], x[1], y[0], y[1]]
    else:
        parts = [x, y]
    return b"".join(int(p).to_bytes(fb, "little") for p in parts)


def points_to_wire(curve, pts):
    """list of oracle Points -> uint8 [n, POINT_BYTES] (infinity -> all zero)."""
    pb = POINT_BYTES[curve]
    out = np.zeros((len(pts), pb), dtype=np.uint8)
    for i, p in enumerate(pts):
        out[i] = np.frombuffer(affine_to_wire(curve, p.toAffine()), dtype=np.uint8)
    return out


def wire_to_affine(curve, row):
    fb = FIELD_BYTES[curve]
    vals = le_to_ints(np.asarray(row, dtype=np.uint8).reshape(-1, fb), fb)
    if curve == BLS12_381_G2:
        return ((vals[0], vals[1]), (vals[2], vals[3]))
    return (vals[0], vals[1])


def scalars_to_wire(scalars):
    return ints_to_le(scalars, 32)
