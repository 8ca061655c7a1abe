"""TEST INFRASTRUCTURE - CPU restatement of the reference's ECDSA verification on secp256k1
(src/abstract/weierstrass.ts:1571-1620 `verify`, Signature :1343-1420, bits2int / bits2int_modN :1439-1455,
DER :215-330 for the Wycheproof vectors).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this; the product path never does.  Pinned by tests/test_oracle_golden.py against the reference's own
vectors (test/vectors/secp256k1/ecdsa.json, test/vectors/wycheproof/ecdsa_secp256k1_* via tests/golden)."""
import hashlib

from .curves import SECP256K1_N, Secp256k1
from .weierstrass import sec1_decode

N = SECP256K1_N


def bits2int(data):
    """weierstrass.ts:1439-1450: leftmost Fn.BITS bits of the byte string."""
    num = int.from_bytes(bytes(data), "big")
    delta = len(data) * 8 - 256
    return num >> delta if delta > 0 else num


def der_parse_int(data):
    """DER._int.decode (weierstrass.ts:287-304): (value, rest)."""
    if len(data) < 2 or data[0] != 0x02:
        raise ValueError("invalid signature integer tag")
    ln = data[1]
    if ln & 0x80:                      # long-form lengths
        nb = ln & 0x7F
        if nb == 0 or nb > 4 or len(data) < 2 + nb:
            raise ValueError("invalid signature integer: bad length")
        if data[2] == 0:
            raise ValueError("tlv.decode(long): zero leftmost byte")
        ln = int.from_bytes(data[2:2 + nb], "big")
        if ln < 128:
            raise ValueError("tlv.decode(long): not minimal encoding")
        off = 2 + nb
    else:
        off = 2
    body = data[off:off + ln]
    if len(body) != ln or ln == 0:
        raise ValueError("invalid signature integer: wrong length")
    if body[0] & 0x80:
        raise ValueError("invalid signature integer: negative")
    if body[0] == 0x00 and ln > 1 and not (body[1] & 0x80):
        raise ValueError("invalid signature integer: unnecessary leading zero")
    return int.from_bytes(body, "big"), data[off + ln:]


def der_to_rs(sig):
    """DER.toSig (weierstrass.ts:308-323)."""
    sig = bytes(sig)
    if len(sig) < 2 or sig[0] != 0x30:
        raise ValueError("invalid signature tag")
    ln = sig[1]
    off = 2
    if ln & 0x80:
        nb = ln & 0x7F
        if nb == 0 or nb > 4 or len(sig) < 2 + nb or sig[2] == 0:
            raise ValueError("bad length")
        ln = int.from_bytes(sig[2:2 + nb], "big")
        if ln < 128:
            raise ValueError("not minimal")
        off = 2 + nb
    body = sig[off:]
    if len(body) != ln:
        raise ValueError("invalid signature: left bytes after parsing")
    r, rest = der_parse_int(body)
    s, rest = der_parse_int(rest)
    if rest:
        raise ValueError("invalid signature: left bytes after parsing")
    return r, s


def verify(sig, message, public_key, lowS=True, prehash=True, fmt="compact"):
    """weierstrass.ts:1583-1620.  sig: 64-byte compact (r || s) or DER; returns bool."""
    msg = hashlib.sha256(bytes(message)).digest() if prehash else bytes(message)
    try:
        if fmt == "der":
            r, s = der_to_rs(sig)
        else:
            sig = bytes(sig)
            if len(sig) != 64:
                raise ValueError("bad signature length")
            r, s = int.from_bytes(sig[:32], "big"), int.from_bytes(sig[32:], "big")
        if not (1 <= r < N and 1 <= s < N):
            raise ValueError("invalid signature: out of range")
        P = sec1_decode(Secp256k1, public_key)
        if P.is0():
            return False
        if lowS and s > (N >> 1):
            return False
        h = bits2int(msg) % N
        is_ = pow(s, -1, N)
        u1, u2 = h * is_ % N, r * is_ % N
        R = Secp256k1.BASE.mulAddUnsafe(u1, P, u2)
        if R.is0():
            return False
        return R.toAffine()[0] % N == r
    except ValueError:
        return False


# ---- BIP-340 Schnorr (src/secp256k1.ts:129-137 taggedHash, :158-170 lift_x, :176-178 challenge, :228-258 verify)
def tagged_hash(tag, *msgs):
    t = hashlib.sha256(tag.encode()).digest()
    return hashlib.sha256(t + t + b"".join(bytes(m) for m in msgs)).digest()


def lift_x(x):
    from .curves import SECP256K1_P as P
    if not (0 < x < P):
        raise ValueError("invalid x: Fail if x >= p")
    c = (x * x * x + 7) % P
    y = pow(c, (P + 1) // 4, P)
    if y * y % P != c:
        raise ValueError("Cannot find square root")
    if y & 1:
        y = P - y
    return Secp256k1.fromAffine((x, y))


def schnorr_verify(sig, message, public_key):
    from .curves import SECP256K1_P as P
    sig, message, public_key = bytes(sig), bytes(message), bytes(public_key)
    if len(sig) != 64 or len(public_key) != 32:
        raise ValueError("expected 64-byte signature and 32-byte public key")
    try:
        Pt = lift_x(int.from_bytes(public_key, "big"))
        r = int.from_bytes(sig[:32], "big")
        if not (0 < r < P):
            return False
        s = int.from_bytes(sig[32:], "big")
        if not (0 < s < N):
            return False
        e = int.from_bytes(tagged_hash("BIP0340/challenge", sig[:32], Pt.toAffine()[0].to_bytes(32, "big"), message), "big") % N
        R = Secp256k1.BASE.mulAddUnsafe(s, Pt, (N - e) % N)
        if R.is0():
            return False
        x, y = R.toAffine()
        return (y & 1) == 0 and x == r
    except ValueError:
        return False


def recover_public_key(sig65, message, prehash=True):
    """Signature.fromBytes(sig, 'recovered').recoverPublicKey (weierstrass.ts:1352-1366, :1391-1407, :1621-1630):
    sig65 = recid || r || s; returns the point Q or raises ValueError where the reference throws."""
    from .curves import SECP256K1_P as P
    sig65 = bytes(sig65)
    if len(sig65) != 65:
        raise ValueError("bad recovered signature length")
    msg = hashlib.sha256(bytes(message)).digest() if prehash else bytes(message)
    rec = sig65[0]
    r, s = int.from_bytes(sig65[1:33], "big"), int.from_bytes(sig65[33:], "big")
    if not (1 <= r < N and 1 <= s < N):
        raise ValueError("invalid signature: out of range")
    if rec not in (0, 1, 2, 3):
        raise ValueError("invalid recovery id")
    radj = r + N if rec in (2, 3) else r
    if not radj < P:
        raise ValueError("invalid recovery id: sig.r+curve.n != R.x")
    R = sec1_decode(Secp256k1, bytes([2 if rec & 1 == 0 else 3]) + radj.to_bytes(32, "big"))
    ir = pow(radj, -1, N)
    h = bits2int(msg) % N
    Q = Secp256k1.BASE.mulAddUnsafe(-h * ir % N, R, s * ir % N)
    if Q.is0():
        raise ValueError("invalid recovery: point at infinify")
    return Q
