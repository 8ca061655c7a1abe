"""secp256k1 ECDSA host shim: `verify` / `verify_batch` with the reference's semantics
(src/abstract/weierstrass.ts:1571-1620, options :1196-1230: lowS true, prehash true, format 'compact').

The shim does what the reference does on the host around the group operation - argument checks, SHA-256 of the
message when `prehash`, bits2int, DER parsing for format 'der', the length / prefix / on-curve check of an
uncompressed key - and hands fixed-size rows (r || s, h, compressed key) to the library; key decompression,
s^-1 mod n, u1 G + u2 P and the comparison run in HIP kernels (`ncg_ecdsa_verify_batch`).
"""
import hashlib

import numpy as np

from ._native import get_engine

P = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEFFFFFC2F
N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141


def _abytes(b, title):
    if not isinstance(b, (bytes, bytearray, memoryview, np.ndarray)):
        raise TypeError('"%s" expected Uint8Array, got type=%s' % (title, type(b).__name__))
    return bytes(b)


def bits2int(data):
    """weierstrass.ts:1439-1450: the leftmost 256 bits of the byte string as an integer."""
    if len(data) > 8192:
        raise ValueError("input is too large")
    num = int.from_bytes(data, "big")
    delta = len(data) * 8 - 256
    return num >> delta if delta > 0 else num


def _der_len(buf, pos):
    """Definite length at buf[pos:]: (length, next position); long form must be minimal (DER._tlv.decode)."""
    if pos >= len(buf):
        raise ValueError("tlv.decode: wrong value length")
    first = buf[pos]
    pos += 1
    if not first & 0x80:
        return first, pos
    nb = first & 0x7F
    if nb == 0:
        raise ValueError("tlv.decode(long): indefinite length not supported")
    if nb > 4:
        raise ValueError("tlv.decode(long): byte length is too big")
    chunk = buf[pos:pos + nb]
    if len(chunk) != nb:
        raise ValueError("tlv.decode: length bytes not complete")
    if chunk[0] == 0:
        raise ValueError("tlv.decode(long): zero leftmost byte")
    length = int.from_bytes(chunk, "big")
    if length < 128:
        raise ValueError("tlv.decode(long): not minimal encoding")
    return length, pos + nb


def _der_int(buf, pos):
    if pos >= len(buf) or buf[pos] != 0x02:
        raise ValueError("tlv.decode: wrong tlv")
    length, pos = _der_len(buf, pos + 1)
    body = buf[pos:pos + length]
    if len(body) != length or length == 0:
        raise ValueError("tlv.decode: wrong value length")
    if body[0] & 0x80:
        raise ValueError("invalid signature integer: negative")
    if body[0] == 0 and length > 1 and not body[1] & 0x80:
        raise ValueError("invalid signature integer: unnecessary leading zero")
    return int.from_bytes(body, "big"), pos + length


def der_to_rs(sig):
    """(r, s) of a DER signature, strict like DER.toSig (weierstrass.ts:308-323)."""
    sig = bytes(sig)
    if not sig or sig[0] != 0x30:
        raise ValueError("tlv.decode: wrong tlv")
    length, pos = _der_len(sig, 1)
    if len(sig) - pos != length:
        raise ValueError("invalid signature: left bytes after parsing")
    r, pos = _der_int(sig, pos)
    s, pos = _der_int(sig, pos)
    if pos != len(sig):
        raise ValueError("invalid signature: left bytes after parsing")
    return r, s


def _compressed_key(pk):
    """33-byte compressed form of a SEC1 key, or None where Point.fromBytes throws before any field work
    (wrong length / prefix, coordinates out of range, uncompressed point off the curve; weierstrass.ts:566-605)."""
    if len(pk) == 33 and pk[0] in (2, 3):
        return pk                                  # x range and the square root are checked on the device
    if len(pk) == 65 and pk[0] == 4:
        x, y = int.from_bytes(pk[1:33], "big"), int.from_bytes(pk[33:], "big")
        if not (0 <= x < P and 0 <= y < P) or (y * y - x * x * x - 7) % P:
            return None
        return bytes([2 + (y & 1)]) + pk[1:33]
    return None


def verify_batch(signatures, messages, publicKeys, lowS=True, prehash=True, format="compact", engine=None, hash_on_device=True):
    """[secp256k1.verify(sig, msg, key, {lowS, prehash, format}) for each triple] in one launch.  With prehash (the
    reference's default) and hash_on_device the SHA-256 of every message runs in a HIP kernel too
    (ncg_ecdsa_verify_batch_msgs); hash_on_device=False hashes here with hashlib."""
    n = len(signatures)
    if len(messages) != n or len(publicKeys) != n:
        raise ValueError("arrays of signatures, messages and public keys must have equal length")
    for name, v in (("lowS", lowS), ("prehash", prehash)):
        if not isinstance(v, bool):
            raise TypeError('"%s" expected boolean' % name)
    if format not in ("compact", "der"):
        raise ValueError('Signature format must be "compact" or "der"')   # 'recovered' is not offered in batch
    S = np.zeros((n, 64), np.uint8)
    H = np.zeros((n, 32), np.uint8)
    # all keys uncompressed (65 bytes, e.g. Ethereum-style callers): the rows go up as they are and the prefix /
    # range / curve-equation checks run on the device; otherwise every key is brought to the 33-byte form here
    all_unc = n > 0 and all(isinstance(k, (bytes, bytearray, memoryview, np.ndarray)) and len(k) == 65 for k in publicKeys)
    K = np.zeros((n, 65 if all_unc else 33), np.uint8)
    live = np.zeros((n,), bool)
    raw_msgs = []
    for i in range(n):
        pk = _abytes(publicKeys[i], "publicKey")
        msg = _abytes(messages[i], "message")
        sig = signatures[i]
        if not isinstance(sig, (bytes, bytearray, memoryview, np.ndarray)):
            raise TypeError("verify expects Uint8Array signature")
        sig = bytes(sig)
        if format == "compact" and len(sig) != 64:                         # validateSigLength: loud, like the reference
            raise ValueError('"signature" expected Uint8Array of length 64, got length=%d' % len(sig))
        dev_hash = prehash and hash_on_device
        if prehash and not dev_hash:
            msg = hashlib.sha256(msg).digest()
        try:
            if format == "der":
                r, s = der_to_rs(sig)
                if not (1 <= r < N and 1 <= s < N):
                    raw_msgs.append(b"")
                    continue
                sig = r.to_bytes(32, "big") + s.to_bytes(32, "big")
            key = pk if all_unc else _compressed_key(pk)
            if key is None:
                raw_msgs.append(b"")
                continue
            h = 0 if dev_hash else bits2int(msg)
        except ValueError:
            raw_msgs.append(b"")
            continue                                                       # the reference's catch: false
        S[i] = np.frombuffer(sig, np.uint8)
        H[i] = np.frombuffer(h.to_bytes(32, "big"), np.uint8)
        K[i] = np.frombuffer(key, np.uint8)
        raw_msgs.append(msg)
        live[i] = True
    if n == 0:
        return []
    if not all_unc:
        K[~live, 0] = 2                                                    # well-formed filler rows; verdict forced below
    eng = engine or get_engine()
    ok = eng.ecdsa_verify_batch_msgs(S, raw_msgs, K, lowS) if (prehash and hash_on_device) else eng.ecdsa_verify_batch(S, H, K, lowS)
    return [bool(a and b) for a, b in zip(ok, live)]


def verify(signature, message, publicKey, **opts):
    return verify_batch([signature], [message], [publicKey], **opts)[0]


def recoverPublicKeyBatch(signatures, messages, prehash=True, isCompressed=True, engine=None):
    """[secp256k1.recoverPublicKey(sig, msg, {prehash}) for each pair] (weierstrass.ts:1621-1630): signatures in the
    'recovered' format (65 bytes: recovery id || r || s), result = the key's SEC1 bytes.  Entries for which the
    reference throws (bad recovery id, r / s out of range, no such R, Q = O) come back as None."""
    n = len(signatures)
    if len(messages) != n:
        raise ValueError("arrays of signatures and messages must have equal length")
    if not isinstance(prehash, bool):
        raise TypeError('"prehash" expected boolean')
    if n == 0:
        return []
    S = np.zeros((n, 65), np.uint8)
    H = np.zeros((n, 32), np.uint8)
    for i in range(n):
        sig = _abytes(signatures[i], "signature")
        if len(sig) != 65:
            raise ValueError('"signature" expected Uint8Array of length 65, got length=%d' % len(sig))
        msg = _abytes(messages[i], "message")
        if prehash:
            msg = hashlib.sha256(msg).digest()
        S[i] = np.frombuffer(sig, np.uint8)
        H[i] = np.frombuffer(bits2int(msg).to_bytes(32, "big"), np.uint8)
    eng = engine or get_engine()
    pts, ok = eng.ecdsa_recover_batch(S, H)
    enc, enc_ok = eng.encode_points_batch(0, pts)                          # SEC1 compressed (curve id 0 = secp256k1)
    out = []
    for i in range(n):
        if not (ok[i] and enc_ok[i]):
            out.append(None)
        elif isCompressed:
            out.append(bytes(enc[i]))
        else:
            out.append(b"\x04" + bytes(pts[i][31::-1]) + bytes(pts[i][:31:-1]))
    return out


def getSharedSecretBatch(secretKeys, publicKeys, isCompressed=True, engine=None):
    """[secp256k1.getSharedSecret(sk, pk, isCompressed) for each pair] (weierstrass.ts:1198-1210): the point
    s * Point.fromBytes(pk) as SEC1 bytes.  Secret keys: 32 big-endian bytes in [1, n) (Fn.fromBytes +
    isValidNot0, else ValueError like the reference); a public key the reference's decoder rejects raises too.
    NOT constant-time: the reference's getSharedSecret multiplies the secret key with the constant-time
    Point.multiply (weierstrass.ts:1198-1210); this batch form runs the GPU's variable-time GLV ladder
    (zero digits skipped, table index taken from the scalar digits), so its timing and memory-access pattern
    depend on the secret scalars.  Use it only where that side channel is acceptable (DESIGN.md section 8:
    the GPU path offers no constant-time multiplication); the reference's own getSharedSecret stays the
    constant-time route.
    """
    from . import curve as G
    K1 = G.secp256k1_Point
    n = len(secretKeys)
    if len(publicKeys) != n:
        raise ValueError("arrays of secret keys and public keys must have equal length")
    ks = []
    for sk in secretKeys:
        sk = _abytes(sk, "secretKey")
        if len(sk) != 32:
            raise ValueError('"secretKey" expected Uint8Array of length 32, got length=%d' % len(sk))
        k = int.from_bytes(sk, "big")
        if not (1 <= k < N):
            raise ValueError("invalid private key")
        ks.append(k)
    comp = []
    for i, pk in enumerate(publicKeys):
        c = _compressed_key(_abytes(pk, "publicKey"))
        if c is None:
            raise ValueError("bad point: invalid public key at index %d" % i)
        comp.append(c)
    if n == 0:
        return []
    pts = G.fromBytesBatch(K1, comp, engine=engine)
    for i, p in enumerate(pts):
        if p is None:
            raise ValueError("bad point: invalid public key at index %d" % i)
    shared = G.multiplyUnsafeBatch(K1, pts, ks, engine=engine)      # k in [1, n) and P != O: never ZERO
    if isCompressed:
        return [bytes(b) for b in G.toBytesBatch(K1, shared, engine=engine)]
    return [b"\x04" + x.to_bytes(32, "big") + y.to_bytes(32, "big") for x, y in (p.toAffine() for p in shared)]
