"""secp256k1 BIP-340 Schnorr host shim: `verify` / `verify_batch` with the reference's semantics
(src/secp256k1.ts:228-258).  The shim checks the argument types and lengths like the reference (loudly), computes
the challenge e = int(taggedHash('BIP0340/challenge', r || pk || m)) mod n with hashlib (the reference hashes on
the host as well, :129-137, :176-178) and hands (sig, e, pk) to the library; the range checks on r and s, lift_x,
R = s G - e P and the three acceptance tests run in HIP kernels (`ncg_schnorr_verify_batch`).
"""
import hashlib

import numpy as np

from ._native import get_engine

N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
_TAG = hashlib.sha256(b"BIP0340/challenge").digest() * 2


def _abytes(b, length, title):
    if not isinstance(b, (bytes, bytearray, memoryview, np.ndarray)):
        raise TypeError('"%s" expected Uint8Array, got type=%s' % (title, type(b).__name__))
    b = bytes(b)
    if length is not None and len(b) != length:
        raise ValueError('"%s" expected Uint8Array of length %d, got length=%d' % (title, length, len(b)))
    return b


def challenge(r_bytes, pk_bytes, msg):
    return int.from_bytes(hashlib.sha256(_TAG + r_bytes + pk_bytes + msg).digest(), "big") % N


def verify_batch(signatures, messages, publicKeys, engine=None, hash_on_device=True):
    """[schnorr.verify(sig, msg, pk) for each triple] in one launch.  hash_on_device: the tagged challenge hash runs in
    a HIP kernel as well (ncg_schnorr_verify_batch_msgs); False computes it here with hashlib."""
    n = len(signatures)
    if len(messages) != n or len(publicKeys) != n:
        raise ValueError("arrays of signatures, messages and public keys must have equal length")
    if n == 0:
        return []
    S = np.zeros((n, 64), np.uint8)
    E = np.zeros((n, 32), np.uint8)
    K = np.zeros((n, 32), np.uint8)
    ms = []
    for i in range(n):
        sig = _abytes(signatures[i], 64, "signature")
        msg = _abytes(messages[i], None, "message")
        pk = _abytes(publicKeys[i], 32, "publicKey")
        S[i] = np.frombuffer(sig, np.uint8)
        K[i] = np.frombuffer(pk, np.uint8)
        # pointToBytes(lift_x(pk)) is pk itself whenever lift_x succeeds; where it fails the verdict is false anyway
        if hash_on_device:
            ms.append(msg)
        else:
            E[i] = np.frombuffer(challenge(sig[:32], pk, msg).to_bytes(32, "big"), np.uint8)
    eng = engine or get_engine()
    if hash_on_device:
        return [bool(x) for x in eng.schnorr_verify_batch_msgs(S, ms, K)]
    return [bool(x) for x in eng.schnorr_verify_batch(S, E, K)]


def verify(signature, message, publicKey, engine=None, hash_on_device=True):
    return verify_batch([signature], [message], [publicKey], engine, hash_on_device)[0]
