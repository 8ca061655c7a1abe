"""Python side of oracle/ref_js: build the type-stripped copy of the reference (when /root/reference is present) and run the
REFERENCE's own TypeScript code on this machine's Node for known answers and CPU timings.  TEST INFRASTRUCTURE: only tests/,
__graft_entry__ and bench.py's cpu_baseline leg use it.  The build artefact oracle/_ref/refjs.bundle is git-ignored (no reference
source is committed); it travels to the GPU box with the snapshot, where /root/reference does not exist."""
import json
import os
import shutil
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
BUNDLE = os.path.join(HERE, "_ref", "refjs.bundle")     # the build artefact: gzip tar of the type-stripped modules (git-ignored)
SRC = os.path.join(HERE, "ref_js")
CURVE_NAME = {0: "secp256k1", 1: "ed25519", 2: "bls12_381_g1", 3: "bls12_381_g2"}
_unpacked = None


def node():
    return shutil.which("node") or shutil.which("nodejs")


def build(reference="/root/reference/src"):
    """oracle/_ref/refjs.bundle from the reference's sources where they lie (no-op when they are absent and a bundle exists).
    The type-stripped modules exist only inside the bundle and, while a test or the bench runs, in a temporary directory:
    no look-alike of a reference source file is left in the working tree."""
    global _unpacked
    if not os.path.isdir(reference):
        return available()
    import tarfile
    with tempfile.TemporaryDirectory() as d:
        subprocess.check_call(["python3", os.path.join(SRC, "downlevel.py"), "--src", reference, "--out", d], stdout=subprocess.DEVNULL)
        for f in ("hashes_shim.mjs", "polyfill.mjs", "run_ref.mjs"):
            shutil.copy(os.path.join(SRC, f), os.path.join(d, f))
        os.makedirs(os.path.dirname(BUNDLE), exist_ok=True)
        with tempfile.TemporaryDirectory() as dh:
            _build_hooked(os.path.dirname(reference.rstrip("/")), dh)
            with tarfile.open(BUNDLE + ".tmp", "w:gz") as tar:
                tar.add(d, arcname="js")
                tar.add(dh, arcname="js_hooked")
        os.replace(BUNDLE + ".tmp", BUNDLE)
    old = os.path.join(HERE, "_ref", "js")
    if os.path.isdir(old):
        shutil.rmtree(old, ignore_errors=True)
    _unpacked = None
    return available()


# ---- the REDIRECTED copy (js_hooked/): the reference with the MSM-backend patch of INTEGRATION.md applied to src/abstract/curve.ts
# (downlevel.py --gpu-hook), laid out like the reference's repository (src/, test/, benchmark/) together with the reference's OWN
# test and benchmark files for the path and the data files they read, so that tests/test_node_redirect.py can run them with the GPU
# underneath.  harness/ holds stand-ins for the test runner / property generator / benchmark loop packages that are not installed.
HOOKED_SRC = ["utils.ts", "abstract/modular.ts", "abstract/curve.ts", "abstract/weierstrass.ts", "abstract/der.ts", "abstract/edwards.ts",
              "abstract/hash-to-curve.ts", "abstract/tower.ts", "abstract/bls.ts", "abstract/fft.ts", "abstract/montgomery.ts",
              "abstract/frost.ts", "abstract/oprf.ts", "secp256k1.ts", "ed25519.ts", "bls12-381.ts", "bn254.ts", "ed448.ts", "misc.ts", "nist.ts"]
HOOKED_TESTS = ["test/point.test.ts", "test/point.helpers.ts", "test/_more-curves.helpers.ts", "test/utils.helpers.ts", "test/utils.ts",
                "benchmark/msm_timings.ts", "benchmark/bls12-381.ts"]
HOOKED_DATA = ["test/vectors/curves-init.json", "test/vectors/ed25519/vectors.txt", "test/vectors/bls12-381/bls12-381-g2-test-vectors.txt"]


def _build_hooked(ref_root, out):
    files = ["src/" + f for f in HOOKED_SRC] + HOOKED_TESTS
    subprocess.check_call(["python3", os.path.join(SRC, "downlevel.py"), "--src", ref_root, "--out", out, "--gpu-hook"] + files,
                          stdout=subprocess.DEVNULL)
    for f in ("hashes_shim.mjs", "polyfill.mjs"):
        shutil.copy(os.path.join(SRC, f), os.path.join(out, f))
    shutil.copytree(os.path.join(SRC, "harness"), os.path.join(out, "harness"))
    for rel in HOOKED_DATA:                      # data files the reference's tests / benchmarks read (fixtures, not source)
        dst = os.path.join(out, rel)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        shutil.copy(os.path.join(ref_root, rel), dst)
        os.chmod(dst, 0o644)
        if rel.endswith(".json"):                # JSON module for Node 12 (downlevel.py rewrites the import)
            with open(dst) as f, open(dst + ".mjs", "w") as g:
                g.write("export default " + json.dumps(json.load(f)) + ";\n")


def available():
    return node() is not None and os.path.exists(BUNDLE)


def hooked_dir():
    """the redirected copy of the bundle (js_hooked/), or None when the bundle predates it"""
    d = os.path.join(os.path.dirname(ref_dir()), "js_hooked")
    return d if os.path.isdir(d) else None


def ref_dir():
    """the bundle unpacked into a per-process temporary directory (removed at exit)"""
    global _unpacked
    if _unpacked is None:
        import atexit
        import tarfile
        d = tempfile.mkdtemp(prefix="ncg_refjs_")
        with tarfile.open(BUNDLE, "r:gz") as tar:
            tar.extractall(d)
        atexit.register(shutil.rmtree, d, True)
        _unpacked = os.path.join(d, "js")
    return _unpacked


def _run(cmd, data, out_bytes, *args, timeout=600):
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "in.bin"), os.path.join(d, "out.bin")
        with open(fin, "wb") as f:
            f.write(data)
        r = subprocess.run([node(), os.path.join(ref_dir(), "run_ref.mjs"), cmd, fin, fout] + [str(a) for a in args],
                           capture_output=True, text=True, timeout=timeout)
        if r.returncode != 0:
            raise RuntimeError("reference run failed: " + (r.stderr or r.stdout)[-600:])
        info = json.loads(r.stdout.strip().splitlines()[-1])
        out = np.fromfile(fout, dtype=np.uint8) if out_bytes else None
    return out, info


def multiply(curve, points_wire, scalars_wire, unsafe=True):
    """rows of Point.multiplyUnsafe / Point.multiply through the reference; wire arrays as in include/ncg.h"""
    p = np.ascontiguousarray(points_wire, dtype=np.uint8)
    s = np.ascontiguousarray(scalars_wire, dtype=np.uint8).reshape(-1, 32)
    p = p.reshape(s.shape[0], -1)
    out, info = _run("mul_unsafe" if unsafe else "mul", np.concatenate([p, s], axis=1).tobytes(), True, CURVE_NAME[curve])
    return out.reshape(p.shape), info


def pippenger(curve, points_wire, scalars_wire):
    p = np.ascontiguousarray(points_wire, dtype=np.uint8)
    s = np.ascontiguousarray(scalars_wire, dtype=np.uint8).reshape(-1, 32)
    out, info = _run("pippenger", p.tobytes() + s.tobytes(), True, CURVE_NAME[curve], timeout=1800)
    return out, info


def ed25519_verify(sigs, msgs, pks, zip215=True):
    rec = bytearray()
    for sg, m, pk in zip(sigs, msgs, pks):
        assert len(m) <= 64
        rec += bytes(sg) + bytes(pk) + len(m).to_bytes(4, "little") + bytes(m).ljust(64, b"\0")
    out, info = _run("ed25519_verify", bytes(rec), True, "zip215" if zip215 else "strict")
    return out.astype(bool), info


def hash_to_curve(curve, msgs, dst, encode=False):
    """bls12_381.G1 / G2 hashToCurve (encode: encodeToCurve) of each message through the reference; returns wire points [n, PB]."""
    rec = bytearray()
    for m in msgs:
        assert len(m) <= 64
        rec += len(m).to_bytes(4, "little") + bytes(m).ljust(64, b"\0")
    g = "g2" if CURVE_NAME[curve].endswith("g2") else "g1"
    out, info = _run("h2c", bytes(rec), True, g, bytes(dst).hex(), *(["encode"] if encode else []))
    return out.reshape(len(msgs), -1), info


def fft(values, inverse=False, brp_in=False, brp_out=False):
    """FFT(rootsOfUnity(Fr), Fr).direct / .inverse of a list of residues of the bls12-381 scalar field (fft.ts:518-577)."""
    data = b"".join(int(v).to_bytes(32, "little") for v in values)
    out, info = _run("fft", data, True, "inverse" if inverse else "direct", int(brp_in), int(brp_out))
    b = out.tobytes()
    return [int.from_bytes(b[i * 32:(i + 1) * 32], "little") for i in range(len(values))], info


def codec(curve, points_wire):
    """Point.toBytes(compressed) of each wire point (and Point.fromBytes of the result, which must give the point back)."""
    p = np.ascontiguousarray(points_wire, dtype=np.uint8)
    out, info = _run("codec", p.tobytes(), True, CURVE_NAME[curve])
    return out.reshape(p.shape[0], -1), info


def point_bench(seconds=2.0):
    r = subprocess.run([node(), os.path.join(ref_dir(), "run_ref.mjs"), "point_bench", str(seconds)], capture_output=True, text=True, timeout=600)
    if r.returncode != 0:
        raise RuntimeError("reference run failed: " + (r.stderr or r.stdout)[-600:])
    return json.loads(r.stdout.strip().splitlines()[-1])
