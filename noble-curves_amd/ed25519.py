"""ed25519 host shim: `verify` / `verify_batch` with the reference's semantics
(src/abstract/edwards.ts:942-989, ZIP-215 default from src/ed25519.ts:162-172).

The shim does the argument checks of the reference and hands (sig, pk, msg) to the library: the
SHA-512 challenge k = SHA-512(R || A || M) mod L (edwards.ts:984, :900-906, :866-868; @noble/hashes in
the reference) and the curve arithmetic both run in HIP kernels (`ncg_ed25519_verify_batch_msgs`).
`hash_on_device=False` computes k with hashlib instead and uses the k32 entry point.
"""
import hashlib

import numpy as np

from ._native import get_engine

L = 0x1000000000000000000000000000000014DEF9DEA2F79CD65812631A5CF5D3ED


def _abytes(b, length, title):
    if not isinstance(b, (bytes, bytearray, memoryview, np.ndarray)):
        raise TypeError('"%s" expected Uint8Array, got type=%s' % (title, type(b).__name__))
    b = bytes(b)
    if length is not None and len(b) != length:
        raise ValueError('"%s" expected Uint8Array of length %d, got length=%d' % (title, length, len(b)))
    return b


def challenge(r_bytes, pk_bytes, msg):
    """hashDomainToScalar + modN_LE (edwards.ts:900-906, :866-868)."""
    return int.from_bytes(hashlib.sha512(bytes(r_bytes) + bytes(pk_bytes) + bytes(msg)).digest(), "little") % L


def verify_batch(sigs, msgs, publicKeys, zip215=True, engine=None, hash_on_device=True):
    """List of booleans, one per (signature, message, publicKey) triple."""
    n = len(sigs)
    if len(msgs) != n or len(publicKeys) != n:
        raise ValueError("arrays of signatures, messages and public keys must have equal length")
    if not isinstance(zip215, bool):
        raise TypeError('"zip215" expected boolean')
    S = np.zeros((n, 64), np.uint8)
    P = np.zeros((n, 32), np.uint8)
    K = np.zeros((n, 32), np.uint8)
    ms = []
    for i in range(n):
        sig = _abytes(sigs[i], 64, "signature")
        msg = _abytes(msgs[i], None, "message")
        pk = _abytes(publicKeys[i], 32, "publicKey")
        S[i] = np.frombuffer(sig, np.uint8)
        P[i] = np.frombuffer(pk, np.uint8)
        if hash_on_device:
            ms.append(msg)
        else:
            K[i] = np.frombuffer(challenge(sig[:32], pk, msg).to_bytes(32, "little"), np.uint8)
    eng = engine or get_engine()
    if hash_on_device:
        off = np.zeros((n + 1,), np.uint64)
        if n:
            off[1:] = np.cumsum([len(m) for m in ms], dtype=np.uint64)
        blob = np.frombuffer(b"".join(ms), np.uint8) if n and off[n] else np.zeros((0,), np.uint8)
        return [bool(x) for x in eng.ed25519_verify_batch_msgs(S, P, blob, off, zip215)]
    return [bool(x) for x in eng.ed25519_verify_batch(S, P, K, zip215)]


def verify(sig, msg, publicKey, zip215=True, engine=None):
    """eddsa.verify (edwards.ts:942): one signature = a batch of one."""
    return verify_batch([sig], [msg], [publicKey], zip215, engine)[0]
