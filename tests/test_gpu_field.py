"""Field-level parity ON THE DEVICE: the field code of the kernels - fe9.hpp's
inline-asm multiply-add chains, interleaved fold, bound-typed lazy add / sub / normalise / zero test /
addition-chain inversion for secp256k1 and ed25519, and the radix-2^29 Montgomery code for bls12-381 -
against big-int arithmetic (modular.ts:940-982 values), with operands at the top of what each bound type
admits and at the special values of each prime."""
import numpy as np
import pytest

from noble_curves_amd import get_engine
from oracle.curves import BLS_P, ED25519_P, SECP256K1_P, makeRng

pytestmark = pytest.mark.gpu
U = (1 << 29) + (1 << 19)
M29 = (1 << 29) - 1


def _val(l):
    return sum(int(x) << (29 * i) for i, x in enumerate(l))


def _words(out_row):
    return sum(int(w) << (32 * i) for i, w in enumerate(out_row))


@pytest.mark.parametrize("fid,p", [(0, SECP256K1_P), (1, ED25519_P)])
def test_fe9_ops_on_device(fid, p):
    eng = get_engine()
    rng = makeRng(0xF1E1D + fid)

    def operand(B, kind):
        if kind == 0:
            return [B * U - 1] * 9
        if kind == 1:
            return [0] * 8 + [B * U - 1]
        if kind == 2:
            return [B * U - 1] + [0] * 8
        if kind == 3:                                   # a canonical special value spread over tight limbs
            x = [0, 1, 2, p - 1, p - 2, (p - 1) // 2, p, 2 * p, (1 << 256) - 1][rng.rnd64() % 9]
            return [(x >> (29 * i)) & M29 for i in range(9)]
        return [rng.rndBelow(B * U) if rng.rnd64() % 4 else B * U - 1 for _ in range(9)]

    for variant in (11, 12, 17, 71, 23, 22, 33, 77):
        A, B = variant // 10, variant % 10
        a = np.array([operand(A, k % 6) for k in range(96)], dtype=np.uint32)
        b = np.array([operand(B, (k * 5 + 1) % 7) for k in range(96)], dtype=np.uint32)
        va, vb = [_val(r) for r in a], [_val(r) for r in b]
        got = eng.field_check(fid, 0, variant, a, b)
        assert [_words(r) for r in got] == [x * y % p for x, y in zip(va, vb)], variant
        assert int(eng.field_check(fid, 9, variant, a, b)[:, 0].max()) < U
        if A <= 2:
            assert [_words(r) for r in eng.field_check(fid, 1, variant, a, b)] == [x * x % p for x in va]
        if A + B <= 7:
            assert [_words(r) for r in eng.field_check(fid, 2, variant, a, b)] == [(x + y) % p for x, y in zip(va, vb)]
        if A + B + 1 <= 7:
            assert [_words(r) for r in eng.field_check(fid, 3, variant, a, b)] == [(x - y) % p for x, y in zip(va, vb)]
        if A + 1 <= 7:
            assert [_words(r) for r in eng.field_check(fid, 4, variant, a, b)] == [(-x) % p for x in va]
        assert [_words(r) for r in eng.field_check(fid, 6, variant, a, b)] == [x % p for x in va]
        assert [int(r[0]) for r in eng.field_check(fid, 7, variant, a, b)] == [1 if x % p == 0 else 0 for x in va]
    # exact multiples of p written with loose limbs must test as zero; inversion by the addition chains
    mult = []
    for j in (0, 1, 2, 5, 31, 33, 64, 100):
        x = j * p
        l = [(x >> (29 * i)) & M29 for i in range(8)] + [x >> (29 * 8)]
        if l[8] < 7 * U:
            mult.append(l)
    zl = np.array(mult, dtype=np.uint32)
    assert [int(r[0]) for r in eng.field_check(fid, 7, 71, zl, zl)] == [1] * len(mult)
    xs = [1, 2, 3, p - 1, p - 2, 0] + [rng.rndBelow(p) for _ in range(26)]
    xl = np.array([[(x >> (29 * i)) & M29 for i in range(9)] for x in xs], dtype=np.uint32)
    assert [_words(r) for r in eng.field_check(fid, 5, 11, xl, xl)] == [pow(x, -1, p) if x else 0 for x in xs]


def test_fe29_ops_on_device():
    eng = get_engine()
    rng = makeRng(0xB15)
    p = BLS_P
    xs = [0, 1, 2, p - 1, p - 2, (p - 1) // 2] + [rng.rndBelow(p) for _ in range(58)]
    ys = [xs[(i * 7 + 3) % len(xs)] for i in range(len(xs))]
    w = lambda v: [(v >> (32 * i)) & 0xFFFFFFFF for i in range(12)]  # noqa: E731
    a, b = np.array([w(x) for x in xs], dtype=np.uint32), np.array([w(y) for y in ys], dtype=np.uint32)
    for op, f in ((0, lambda x, y: x * y % p), (1, lambda x, y: x * x % p), (2, lambda x, y: (x + y) % p),
                  (3, lambda x, y: (x - y) % p), (4, lambda x, y: (-x) % p), (5, lambda x, y: pow(x, -1, p) if x else 0)):
        got = eng.field_check(2, op, 0, a, b)
        assert [_words(r) for r in got] == [f(x, y) for x, y in zip(xs, ys)], op
