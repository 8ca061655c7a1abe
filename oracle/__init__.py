"""CPU oracle for the noble-curves hot path.  TEST INFRASTRUCTURE ONLY.

This package restates, in plain Python big-int arithmetic, the algorithms of the
reference's hot path (abstract/modular.ts -> abstract/weierstrass.ts / edwards.ts
-> abstract/curve.ts).  Every function cites the reference file:line it follows.

Nothing under ``oracle/`` is product code: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker.  The shipped path (``noble-curves_amd``)
never imports this package and fails loudly when its HIP library is missing.

Parity status: PINNED.  ``tests/test_oracle_golden.py`` checks this oracle
against the reference's own fixtures (extracted to ``tests/golden/`` by
``tests/golden/make_golden.py``): secp256k1 k*G table, bitcoinjs
pointMultiply/pointAdd vectors, the GLV endomorphism vectors, zkcrypto
bls12-381 G1/G2 i*G tables, cr.yp.to ed25519 sign.input lines and the 196
ZIP-215 verdicts in both modes.
"""
