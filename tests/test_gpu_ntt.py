"""GPU parity for the NTT over bls12-381 Fr through the C ABI (`ncg_ntt`)."""
import numpy as np
import pytest

from noble_curves_amd import fft as G
from noble_curves_amd import get_engine
from noble_curves_amd._native import ints_to_le, le_to_ints
from oracle.curves import Fr_bls, makeRng
from oracle.fft import FFT, RootsOfUnity, bitReversalPermutation

from helpers import load_golden

pytestmark = pytest.mark.gpu
R = Fr_bls.ORDER



def test_fft_known_answers_gpu():
    """test/fft.test.ts:155-183, :221-251 through the mirror (same calls as the reference's test)."""
    kat = load_golden("fft_kat.json")
    roots = G.rootsOfUnity(G.bls12_381_Fr, 7)
    assert roots.roots(3) == [int(x) for x in kat["roots3"]]
    assert roots.brp(3) == [int(x) for x in kat["brp3"]]
    fftFr = G.FFT(roots, G.bls12_381_Fr)
    inp, exp = [int(x) for x in kat["basic_input"]], [int(x) for x in kat["basic_exp"]]
    brp = G.bitReversalPermutation
    assert fftFr.direct(inp) == exp
    assert fftFr.direct(brp(inp), True) == exp
    assert brp(fftFr.direct(inp, False, True)) == exp
    assert brp(fftFr.direct(brp(inp), True, True)) == exp
    assert fftFr.inverse(fftFr.direct(inp)) == inp
    assert fftFr.inverse(fftFr.direct(inp, False, True), True) == inp
    assert brp(fftFr.inverse(fftFr.direct(inp), False, True)) == inp
    assert brp(fftFr.inverse(fftFr.direct(inp, False, True), True, True)) == inp
    assert fftFr.direct([5]) == [5] and fftFr.inverse([5]) == [5]
    with pytest.raises(ValueError, match="FFT: Polynomial size should be power of two"):
        fftFr.inverse([])
    with pytest.raises(ValueError, match="FFT: Polynomial size should be power of two"):
        fftFr.direct([1, 2, 3])
    with pytest.raises(ValueError, match="rootsOfUnity: wrong bits"):
        roots.roots(33)


@pytest.mark.parametrize("bits", [1, 2, 5, 9, 10, 11, 12, 14])
def test_ntt_matches_oracle_all_orderings(bits):
    """every (inverse, brpInput, brpOutput) combination; sizes straddle the 1-pass / 2-pass boundary"""
    rng = makeRng(0x4E77 + bits)
    oroots = RootsOfUnity(Fr_bls, 7)
    of = FFT(oroots, Fr_bls)
    f = G.FFT(G.rootsOfUnity(G.bls12_381_Fr, 7))
    x = [rng.rndBelow(R) for _ in range(1 << bits)]
    x[0], x[1] = 0, R - 1
    for flags in range(8):
        inv, bi, bo = bool(flags & 1), bool(flags & 2), bool(flags & 4)
        exp = (of.inverse if inv else of.direct)(x, bi, bo)
        got = (f.inverse if inv else f.direct)(x, bi, bo)
        assert got == exp, (bits, flags)


def test_ntt_three_pass_size_matches_oracle():
    """2^19 = 10 + 5 + 4 stages: three passes, both butterfly kinds, folded bit reversal through the workspace"""
    bits = 19
    rng = makeRng(0x4E7719)
    oroots = RootsOfUnity(Fr_bls, 7)
    f = G.FFT(G.rootsOfUnity(G.bls12_381_Fr, 7))
    x = [rng.rndBelow(R) for _ in range(1 << bits)]
    # oracle via the defining sum at a few output indices (an O(N) check per index)
    w = oroots.omega(bits)
    y = f.direct(x)
    for k in (0, 1, 2, 12345, (1 << bits) - 1, 1 << 18):
        wk = pow(w, k, R)
        acc, cur = 0, 1
        for xi in x:
            acc = (acc + xi * cur) % R
            cur = cur * wk % R
        assert y[k] == acc, k
    yb = f.direct(x, False, True)
    assert yb == bitReversalPermutation(y)
    assert f.inverse(yb, True) == x
    assert f.inverse(y) == x
    assert f.direct(bitReversalPermutation(x), True) == y


def test_ntt_extreme_values_at_the_bench_size():
    """2^22 (the bench's size: 8 + 7 + 7 stages natural -> natural, 10 + 6 + 6 otherwise) on the inputs that build up the
    largest lazily reduced values the fr29 butterflies can meet - every coefficient r - 1, and r - 1 / 0 alternating - whose
    transforms are known in closed form: N (r - 1) at index 0 (and N/2 (r - 1) at 0 and N/2), zero elsewhere.  Raw arrays
    in and out; natural and bit-reversed output, inverse round trip."""
    eng = get_engine()
    bits = 22
    n = 1 << bits
    roots = G.rootsOfUnity(G.bls12_381_Fr, 7)
    om = roots.omega(bits)
    top = np.frombuffer((R - 1).to_bytes(32, "little"), dtype=np.uint8)
    x = np.tile(top, (n, 1))
    for flags_out in (False, True):
        y = eng.ntt(bits, x, om, brp_output=flags_out)
        assert int.from_bytes(y[0].tobytes(), "little") == n * (R - 1) % R
        assert not y[1:].any()
    x2 = x.copy()
    x2[1::2] = 0
    y = eng.ntt(bits, x2, om)
    half = (n // 2) * (R - 1) % R
    assert int.from_bytes(y[0].tobytes(), "little") == half and int.from_bytes(y[n // 2].tobytes(), "little") == half
    y[0] = 0
    y[n // 2] = 0
    assert not y.any()
    yb = eng.ntt(bits, x2, om, brp_output=True)            # bit-reversed: index N/2 lands at 1
    assert int.from_bytes(yb[0].tobytes(), "little") == half and int.from_bytes(yb[1].tobytes(), "little") == half
    back = eng.ntt(bits, yb, om, inverse=True, brp_input=True)
    assert (back == x2).all()


def test_ntt_batch_and_raw_arrays():
    """a batch of polynomials in one call == the transforms one by one; uint8 arrays pass through"""
    eng = get_engine()
    roots = G.rootsOfUnity(G.bls12_381_Fr, 7)
    rng = makeRng(0xBA7C4)
    bits, batch = 11, 5
    polys = [[rng.rndBelow(R) for _ in range(1 << bits)] for _ in range(batch)]
    data = ints_to_le([v for p in polys for v in p], 32)
    for flags in (0, 1, 4, 7):
        out = eng.ntt(bits, data, roots.omega(bits), inverse=bool(flags & 1), brp_input=bool(flags & 2),
                      brp_output=bool(flags & 4))
        for b in range(batch):
            one = eng.ntt(bits, data[b << bits:(b + 1) << bits], roots.omega(bits), inverse=bool(flags & 1),
                          brp_input=bool(flags & 2), brp_output=bool(flags & 4))
            assert (out[b << bits:(b + 1) << bits] == one).all()
    f = G.FFT(roots)
    raw = f.direct(data[:1 << bits])
    assert isinstance(raw, np.ndarray) and le_to_ints(raw, 32) == f.direct(polys[0])


def test_ntt_rejects_bad_root_and_range():
    eng = get_engine()
    from noble_curves_amd._native import NativeError
    data = ints_to_le([1, 2, 3, 4], 32)
    with pytest.raises(NativeError, match="primitive 2\\^2-th root"):
        eng.ntt(2, data, 5)
    f = G.FFT(G.rootsOfUnity(G.bls12_381_Fr, 7))
    with pytest.raises(ValueError, match="outside of range"):
        f.direct([R, 0])


def test_fft_algebra_properties_gpu():
    """test/fft.test.ts:544-640 'random and algebra properties' through the device transform: round trips,
    additivity, scalar multiples, constant / zero polynomials, the convolution theorem, eval(a*b) = eval(a) eval(b)."""
    rng = makeRng(0xA16EB7A)
    f = G.FFT(G.rootsOfUnity(G.bls12_381_Fr, 7))
    for n in (8, 256, 2048):
        a = [rng.rndBelow(R) for _ in range(n)]
        b = [rng.rndBelow(R) for _ in range(n)]
        c = rng.rndBelow(R - 1) + 1
        assert f.inverse(f.direct(a)) == a and f.direct(f.inverse(a)) == a
        fa, fb = f.direct(a), f.direct(b)
        assert f.direct([(x + y) % R for x, y in zip(a, b)]) == [(x + y) % R for x, y in zip(fa, fb)]
        assert f.direct([x * c % R for x in a]) == [x * c % R for x in fa]
        out = f.direct([c] * n)
        assert out[0] == c * n % R and not any(out[1:])
        assert not any(f.direct([0] * n))
        # convolution theorem on zero-padded halves: direct(a) .* direct(b) = direct(a (*) b)
        h = n // 2
        a0, b0 = a[:h] + [0] * h, b[:h] + [0] * h
        conv = [0] * n
        if n <= 256:
            for i in range(h):
                for j in range(h):
                    conv[i + j] = (conv[i + j] + a0[i] * b0[j]) % R
            assert f.inverse([x * y % R for x, y in zip(f.direct(a0), f.direct(b0))]) == conv
        # eval(a*b)(x) = eval(a)(x) eval(b)(x) through the transform-domain product
        prod = f.inverse([x * y % R for x, y in zip(f.direct(a0), f.direct(b0))])
        x = rng.rndBelow(R)

        def ev(p):
            acc = 0
            for coef in reversed(p):
                acc = (acc * x + coef) % R
            return acc
        assert ev(prod) == ev(a0) * ev(b0) % R


def test_fft_is_evaluation_at_roots_gpu():
    """test/fft.test.ts:642-648 'direct == eval at roots' (and the bit-reversed form)"""
    rng = makeRng(0xD0F7)
    roots = G.rootsOfUnity(G.bls12_381_Fr, 7)
    f = G.FFT(roots)
    for bits in (3, 6):
        n = 1 << bits
        a = [rng.rndBelow(R) for _ in range(n)]
        om = roots.roots(bits)

        def ev(p, x):
            acc = 0
            for coef in reversed(p):
                acc = (acc * x + coef) % R
            return acc
        exp = [ev(a, w) for w in om]
        assert f.direct(a) == exp
        assert f.direct(a, False, True) == G.bitReversalPermutation(exp)
        assert roots.inverse(bits)[1:] == om[1:][::-1] and roots.inverse(bits)[0] == 1
