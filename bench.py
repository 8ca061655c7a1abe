#!/usr/bin/env python3
"""Benchmark of the MI355X hot path (BASELINE.json metric: EC scalar-mults/s and MSM points/s).

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

Primary line (`value`): configs[1] - secp256k1 batch variable-base scalar multiplication
(Point.multiplyUnsafe semantics, GLV), 2^20 (P_i, k_i) pairs per GPU, inputs resident in HBM.
`extra.msm_g1`: configs[3] - bls12-381 G1 Pippenger MSM, 2^20 points per GPU (weak); for N > 1 the points
are sharded and combined inside the C ABI (`ncg_msm_sharded_dev`: one RCCL all-gather of the grouped
window sums + an on-device add); `extra.msm_g1_strong` is ONE 2^20-point MSM split over the N GPUs;
`extra.msm_g2` / `msm_g2_strong` the same for configs[4] (G2, 2^18); `extra.ed25519_verify` configs[2]
with the challenge hash on the device (and the kernel-only rate beside it); `extra.ntt_fr` SURVEY 8f row 3.

A "step" is one pass of the hot path over the whole synthetic batch.  Every result is verified before
any throughput is printed: sampled outputs against the CPU oracle's C restatement, plus full-size
identities (sum of all outputs == (sum k_i (a+i b)) G through the MSM path; MSM == (sum (a+i b) s_i) G,
the construction of the reference's test/slow-curves.test.ts:185-252).

Roofline fields: `roofline.achieved/frac` = algorithmic bytes / event-timed kernel
time vs 8 TB/s (the contract's definition; this path is bound by integer VALU issue, not HBM, so the
fraction is small by nature - SURVEY 8d); `roofline.traffic` = HBM bytes per launch from rocprofv3 PMC
passes, measured live in this run when rocprofv3 is usable (`traffic_source: "live"`), else read from the
committed profile of the same command; `roofline.valu` = executed VALU work: wave-instructions from the
SQ_INSTS_VALU counter and statically counted v_mad_u64_u32 per item, against the measured issue ceilings.
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0                      # MI355X_MICROARCH.md: HBM3E 8 TB/s
N_SIMD, CLOCK_HZ = 256 * 4, 2.4e9
CYC_MAD, CYC_PLAIN = 5.59, 2.89            # measured cycles per wave-instruction and SIMD at 8 waves/SIMD (profiles/r03_valu_rates.jsonl):
                                           # v_mad_u64_u32 / v_add_u32; 64 lanes / 5.59 cycles x 1024 SIMDs x 2.4 GHz = the MAD_PEAK below
MAD_PEAK = 3.08e13                         # v_mad_u64_u32 ceiling measured on MI355X (profiles/r01_ubench_instr_rates.json,
                                           # profiles/r02_valu_rates.jsonl: ~2x the issue time of a plain 32-bit VALU op)
PLAIN_PEAK = N_SIMD * 64 * CLOCK_HZ / CYC_PLAIN   # plain 32-bit VALU ceiling, lane-ops/s (v_add_u32, same measurement)
PMC_PROFILE = os.path.join(ROOT, "profiles", "r06_pmc.json")
if not os.path.exists(PMC_PROFILE):
    PMC_PROFILE = os.path.join(ROOT, "profiles", "r04_pmc.json")
# FETCH_SIZE / WRITE_SIZE -> bytes, calibrated per access pattern with tools/pmc_calib on known byte counts
# (profiles/r04_pmc.json "calibration"): item-major 64 B gathers count at face value, 16 B/lane coalesced
# streams at half (MI355X_MICROARCH.md, HBM), writes at face value.
FETCH_FACTOR = {"gather": 1.0, "stream": 2.0}
KERNELS = {   # workload -> (kernel-name prefix in rocprofv3 output, FETCH access pattern)
    "secp256k1": ("ncg::k_mul_var_gtab<ncg::CurveSecp", "gather"),
    "msm_g1": ("ncg::k_msm_accum<ncg::CurveG1", "gather"),
    "msm_g2": ("ncg::k_msm_accum<ncg::CurveG2P", "gather"),
    "ed25519": ("ncg::k_ed25519_verify", "gather"),
    "ntt": ("ncg::k_ntt_pass", "stream"),
}
KERNEL_FAMILY = {   # workload -> (name prefix, substring, the kernel that runs once per unit): VALU counters summed over the family
    "msm_g1": ("ncg::k_msm_", "<ncg::CurveG1", "ncg::k_msm_tail<"),
    "msm_g2": ("ncg::k_msm_", "<ncg::CurveG2P", "ncg::k_msm_tail<"),
}


# ---------------------------------------------------------------------------- PMC (rocprofv3) plumbing
def _pmc_committed():
    try:
        with open(PMC_PROFILE) as f:
            return json.load(f)["kernels"]
    except (OSError, KeyError, ValueError):
        return {}


def _pmc_lookup(kernels, prefix):
    """Counters of the kernel whose name starts with `prefix`; several instantiations of one template (the NTT
    pass is compiled per tile size) are merged into their launch-weighted average."""
    hits = [v for name, v in kernels.items() if name.startswith(prefix)]
    if len(hits) <= 1:
        return hits[0] if hits else None
    out, wsum = {}, {}
    for v in hits:
        for c, x in v.items():
            if c.startswith("launches_") or c == "_n" or not isinstance(x, (int, float)):
                continue
            w = float(v.get("launches_" + c) or v.get("_n") or 1.0)
            out[c] = out.get(c, 0.0) + x * w
            wsum[c] = wsum.get(c, 0.0) + w
    return {c: out[c] / wsum[c] for c in out}


def pmc_live(workload, log2n, budget_s=420.0):
    """Three rocprofv3 counter passes over a short child run of this script (counters alone with
    --kernel-trace, as MI355X_MICROARCH.md prescribes).  Returns {kernel: {counter: avg per launch}} or None."""
    exe = shutil.which("rocprofv3")
    if not exe:
        return None
    try:
        import pmc_summary
    except ImportError:
        return None
    passes = [["SQ_INSTS_VALU", "SQ_INSTS_VALU_INT64", "SQ_WAVES", "SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE"], ["FETCH_SIZE"], ["WRITE_SIZE"]]
    out = {}
    t0 = time.perf_counter()
    tmp = tempfile.mkdtemp(prefix="ncg_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        for i, ctrs in enumerate(passes):
            left = budget_s - (time.perf_counter() - t0)
            if left < 30:
                return None
            d = os.path.join(tmp, "p%d" % i)
            cmd = [exe, "--pmc"] + ctrs + ["--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable,
                                          os.path.abspath(__file__), "--workload", workload, "--log2n", str(log2n), "--steps", "3",
                                          "--warmup", "1", "--no-cpu-baseline", "--no-live-pmc", "--quick-verify"]
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=left)
            if r.returncode != 0:
                return None
            counters, durs = pmc_summary.read_pass(d)
            for k, cs in counters.items():
                e = out.setdefault(k, {})
                for c, vals in cs.items():
                    e[c] = sum(vals) / len(vals)
                    e["_n"] = len(vals)
                if durs.get(k) and "SQ_INSTS_VALU" in cs:
                    e["avg_ms_valu_pass"] = sum(durs[k]) / len(durs[k]) / 1e6
        return out
    except (subprocess.SubprocessError, OSError, ValueError, KeyError):
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


class Pmc:
    """HBM traffic and executed-VALU figures per kernel: live counters if this run collected them, else the
    committed profile (profiles/r04_pmc.json, same command, earlier box)."""

    def __init__(self, live):
        self.live = live
        self.committed = _pmc_committed()

    def _get(self, prefix):
        if self.live:
            v = _pmc_lookup(self.live, prefix)
            if v:
                return v, "live"
        v = _pmc_lookup(self.committed, prefix)
        return (v, os.path.relpath(PMC_PROFILE, ROOT)) if v else (None, None)

    def traffic(self, workload, launches_per_unit=1):
        prefix, pattern = KERNELS[workload]
        v, src = self._get(prefix)
        if not v or "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
            return None, None
        b = (v["FETCH_SIZE"] * FETCH_FACTOR[pattern] + v["WRITE_SIZE"]) * 1024.0 * launches_per_unit
        return int(b), src

    def valu_counters(self, workload):
        """SQ_INSTS_VALU / SQ_INSTS_VALU_INT64 per unit of the workload: per launch of the dominant kernel, or - the MSMs - summed
        over every kernel of the MSM (accumulate, fix-up, fold levels, tail) and divided by the number of MSMs the pass ran."""
        fam = KERNEL_FAMILY.get(workload)
        if fam:
            for kernels, src in ((self.live, "live"), (self.committed, os.path.relpath(PMC_PROFILE, ROOT))):
                if not kernels:
                    continue
                tot, n_unit = {}, 0.0
                for name, v in kernels.items():
                    if not (name.startswith(fam[0]) and fam[1] in name):
                        continue
                    for c in ("SQ_INSTS_VALU", "SQ_INSTS_VALU_INT64"):
                        if c in v:
                            tot[c] = tot.get(c, 0.0) + v[c] * float(v.get("launches_" + c) or v.get("_n") or 1.0)
                    if name.startswith(fam[2]):
                        n_unit += float(v.get("launches_SQ_INSTS_VALU") or v.get("_n") or 0.0)
                if n_unit and "SQ_INSTS_VALU" in tot:
                    return {c: x / n_unit for c, x in tot.items()}, src + " (all k_msm_* kernels of the curve, per MSM)"
            return None, None
        prefix, _ = KERNELS[workload]
        v, src = self._get(prefix)
        if not v or "SQ_INSTS_VALU" not in v:
            return None, None
        return v, src


BOUND_NOTE = ("bound by VALU issue (integer multiply-add chains), not by HBM or MFMA - SURVEY 8d; `achieved` / `peak` / `frac` "
              "are the contract's HBM figures (algorithmic bytes / kernel time vs 8 TB/s), the fractions that describe the "
              "kernel are roofline.valu.mad_frac and plain_frac")


def _int64_mad_share(prefix):
    """share of v_mad_u64_u32 among the instructions SQ_INSTS_VALU_INT64 counts (v_mad_u64_u32, v_mad_i64_i32, v_lshl_add_u64 -
    profiles/r06_valu_calib.json), from the disassembly of the shipped kernels (tools/int64_share.py -> profiles/r06_int64_mad_share.json)."""
    try:
        with open(os.path.join(ROOT, "profiles", "r06_int64_mad_share.json")) as f:
            tab = json.load(f)["kernels"]
    except (OSError, KeyError, ValueError):
        return 1.0
    hits = [v for k, v in tab.items() if k.startswith(prefix)]
    return min(hits) if hits else 1.0


def valu_block(pmc, workload, kern_s, ref_mac, exec_mads, launches=1):
    """roofline.valu: reference-equivalent work, executed multiplier work (from the SQ_INSTS_VALU_INT64 counter where a counter
    pass exists, cross-checked against the static count of the operation sequence) and the other VALU instructions, each against
    the ceiling measured for its class.  Every *_frac is <= 1 by construction; their sum (pipe_demand_sum) is NOT a fraction."""
    blk = {"ref_equiv_mac_per_s": ref_mac / kern_s, "ref_equiv_note": "limb-MACs of the reference's op sequence (SURVEY 8d), "
           "not executed work", "mad_peak_per_s": MAD_PEAK, "static_mad_per_s": exec_mads / kern_s,
           "static_mad_note": "v_mad_u64_u32 per item counted from the kernel's operation sequence (DESIGN.md section 5) x items"}
    c, src = pmc.valu_counters(workload)
    insts = c.get("SQ_INSTS_VALU") if c else None
    int64 = c.get("SQ_INSTS_VALU_INT64") if c else None
    mads = exec_mads
    blk["mad_source"] = "static count"
    if insts and int64:
        share = _int64_mad_share(KERNELS[workload][0])
        mads = int64 * share * 64.0 * launches
        blk.update({"mad_source": "SQ_INSTS_VALU_INT64 x %.4f (v_mad_u64_u32 share of the 64-bit-class VALU instructions, disassembly) x 64 lanes; %s" % (share, src),
                    "counter_over_static": mads / exec_mads})
    blk["executed_mad_per_s"] = mads / kern_s
    blk["mad_frac"] = mads / kern_s / MAD_PEAK
    if insts:
        plain = max(0.0, insts * launches - mads / 64.0)           # wave-instructions that are not multiply-adds
        blk.update({"plain_valu_per_s": plain * 64.0 / kern_s, "plain_peak_per_s": PLAIN_PEAK, "plain_frac": plain * 64.0 / kern_s / PLAIN_PEAK,
                    "sq_insts_valu_per_launch": insts, "mad_share_of_valu_insts": mads / 64.0 / (insts * launches), "valu_source": src})
        blk["pipe_demand_sum"] = blk["mad_frac"] + blk["plain_frac"]
        blk["pipe_demand_note"] = ("mad_frac + plain_frac: what the two instruction classes would need if nothing overlapped, each at the rate "
                                   "measured for it alone (tools/valu_rates.hip).  NOT a fraction of time: it passes 1 where plain instructions "
                                   "issue in the shadow of a multiply-add of another wave; no gfx950 counter reports VALU busy CYCLES "
                                   "(SQ_ACTIVE_INST_VALU counts one per instruction whatever its pass count, profiles/r06_valu_calib.json)")
    return blk


# ---- statically counted multiplier work (v_mad_u64_u32 per item); see DESIGN.md section 5 for the derivations
FE9_M, FE9_S = 106, 70          # secp256k1 fe9.hpp: 81 products + 18 fold + 7 tail; 45 + 18 + 7
FE29_M, FE29_S = 392, 301       # bls12-381 fp29.hpp: 196 + 196 Montgomery; 105 + 196


def secp_mads_per_mult(W=4, K=16):
    wins = (129 + W - 1) // W                                 # windows per 128-bit half
    dbl, madd = (wins - 1) * W, 2 * wins + 1                   # +1: one parity fix-up on average
    ts = 1 << (W - 1)
    m = dbl * 3 + madd * 8 + (ts - 1) * 8 + (ts - 1) * 4 + wins + 6   # table build, rescale, beta, set-up
    s = dbl * 4 + madd * 3 + (ts - 1) * 3 + (ts - 1) * 1 + 2
    m += 6 + 15.0 / K                                          # batched affine conversion: chain + share of the inversion
    s += 1 + 255.0 / K
    return m * FE9_M + s * FE9_S


def ed25519_mads_per_verify():
    """k_ed25519_verify_half, from its operation sequence (Fe9: 106 multiply-adds per product, 70 per
    square): two decompressions (250 S + 11 M each), 128 shared doublings in 32 windows of 4 (ec_te.hpp: three of four without T,
    4 M + 3 S; the fourth 5 M + 3 S), 32 + 32 additions from the two per-item tables (8 M), 16 + 16 from the shared tables of B and
    2^128 B (7 M), the two 8-entry table builds (7 additions + 8 conversions of 2 M each), three cofactor doublings.  The scalar
    halving is plain operations.  About +-5 %."""
    M, S = FE9_M, FE9_S
    return (2 * (250 * S + 11 * M) + 96 * (4 * M + 3 * S) + 32 * (5 * M + 3 * S) + 64 * 8 * M + 32 * 7 * M
            + 2 * (7 * 8 * M + 8 * 2 * M) + 3 * (5 * M + 3 * S))


def g1_msm_mads_per_point(nwin, n, nb, fused=True):
    # fused: R (Q - X3) - Y1 PPP shares one Montgomery reduction on the unpaired field (fe29.hpp f_mulsub): -196 per add
    save = 196 if fused else 0
    accum = nwin * (8 * FE29_M + 2 * FE29_S - save)                   # one XYZZ mixed add per (point, window)
    fold = nwin * 3.0 * nb * (12 * FE29_M + 2 * FE29_S - save) / n    # fix-up + log-depth fold, ~3 full adds per bucket
    return accum + fold


def make_ed25519_batch(eng, nv, rank, device, stream):
    """configs[2] input: nv (signature, message, public key) triples, every one with its own
    key pair and nonce (A_i = a_i B, R_i = r_i B by the fixed-base batch multiply + batch encoding), 32 random
    message bytes, k_i hashed ON THE DEVICE (sampled against hashlib), s_i = r_i + k_i a_i; 1/64 corrupted (a bit
    of R, of s or of the message); the reference's 196 zip215.json cases replace the tail.  Returns a dict of
    host arrays, device tensors and the expected ZIP-215 verdicts."""
    import hashlib
    from helpers import load_golden
    from noble_curves_amd._native import ED25519
    from oracle.curves import ED25519_L, makeRng
    rng = makeRng(0x6E6F626C6503 + rank)
    g = torch.Generator(device="cpu")
    g.manual_seed(0x6E6F626C6503 + rank)
    # every signature has its own key pair and nonce: A_i = a_i B, R_i = r_i B on the GPU (fixed-base batch
    # multiply + batch encoding), messages = 32 random bytes
    a_sc = gen_scalars(nv, 252, 31 + rank, device)
    r_sc = gen_scalars(nv, 252, 32 + rank, device)
    a_int, r_int = scalars_to_ints(a_sc), scalars_to_ints(r_sc)
    aff = torch.empty((nv, 64), dtype=torch.uint8, device=device)
    inf_t = torch.empty((nv,), dtype=torch.uint8, device=device)
    eng.mul_base_batch_dev(ED25519, nv, dev_ptr(a_sc), dev_ptr(aff), dev_ptr(inf_t), stream)
    torch.cuda.synchronize()
    A_b, okA = eng.encode_points_batch(ED25519, aff.cpu().numpy())
    eng.mul_base_batch_dev(ED25519, nv, dev_ptr(r_sc), dev_ptr(aff), dev_ptr(inf_t), stream)
    torch.cuda.synchronize()
    R_b, okR = eng.encode_points_batch(ED25519, aff.cpu().numpy())
    assert okA.all() and okR.all()
    msgs = torch.randint(0, 256, (nv, 32), dtype=torch.uint8, generator=g).numpy()
    # k_i = SHA-512(R || A || M) mod L on the device (sampled against hashlib below), s_i = r_i + k_i a_i
    sig_np = np.zeros((nv, 64), np.uint8)
    sig_np[:, :32] = R_b
    d_sig0, d_pk, d_msg = (torch.from_numpy(x).to(device) for x in (sig_np, A_b, msgs.reshape(-1)))
    off = torch.arange(0, 32 * (nv + 1), 32, dtype=torch.int64, device=device)
    d_k = torch.empty((nv, 32), dtype=torch.uint8, device=device)
    eng.ed25519_challenge_batch_dev(nv, dev_ptr(d_sig0), dev_ptr(d_pk), dev_ptr(d_msg), dev_ptr(off), dev_ptr(d_k), stream)
    torch.cuda.synchronize()
    k_np = d_k.cpu().numpy()
    k_int = [int.from_bytes(k_np[i].tobytes(), "little") for i in range(nv)]
    for i in list(range(0, 64)) + list(range(64, nv, max(1, nv // 512))):
        assert k_int[i] == int.from_bytes(hashlib.sha512(R_b[i].tobytes() + A_b[i].tobytes() + msgs[i].tobytes()).digest(), "little") % ED25519_L, \
            "device SHA-512 challenge differs from hashlib"
    s_int = [(r + k * a) % ED25519_L for r, k, a in zip(r_int, k_int, a_int)]
    sig_np[:, 32:] = ints_to_le_bytes(s_int)
    expect = np.ones((nv,), bool)
    for i in range(63, nv, 64):                            # 1/64 corrupted: a bit flip in R, s or the message
        kind = (i >> 6) % 3
        if kind == 0:
            sig_np[i, (i >> 8) % 31] ^= 1 << (i % 7)
        elif kind == 1:
            sig_np[i, 33 + (i >> 8) % 20] ^= 1 << (i % 7)
        else:
            msgs[i, (i >> 8) % 32] ^= 1 << (i % 7)
        expect[i] = False
    # the 196 ZIP-215 cases of the reference (small-order / non-canonical encodings) ride along at the end
    zv = load_golden("ed25519_zip215.json")
    nz = len(zv) if nv >= 1024 else 0
    tail = nv - nz
    pk_np = A_b.copy()
    msg_list = [msgs[i].tobytes() for i in range(tail)]
    for j in range(nz):
        sig_np[tail + j] = np.frombuffer(bytes.fromhex(zv[j]["sig_bytes"]), np.uint8)
        pk_np[tail + j] = np.frombuffer(bytes.fromhex(zv[j]["vk_bytes"]), np.uint8)
        msg_list.append(b"Zcash")
        expect[tail + j] = zv[j]["valid_zip215"]
    blob = np.frombuffer(b"".join(msg_list), np.uint8)
    offs = np.zeros((nv + 1,), np.int64)
    offs[1:] = np.cumsum([len(m) for m in msg_list])
    d_sig, d_pk, d_blob, d_off = (torch.from_numpy(x.copy()).to(device) for x in (sig_np, pk_np, blob, offs))
    return {"sig": sig_np, "pk": pk_np, "msgs": msg_list, "expect": expect, "tail": tail, "nz": nz,
            "d_sig": d_sig, "d_pk": d_pk, "d_blob": d_blob, "d_off": d_off}


def ints_to_le_bytes(vals, nbytes=32):
    return np.frombuffer(b"".join(int(v).to_bytes(nbytes, "little") for v in vals), dtype=np.uint8).reshape(-1, nbytes)


def dev_ptr(t):
    return t.data_ptr()


def gen_points(eng, curve, Pt, n, a, b, device, stream):
    """P_i = (a + i b) G, affine wire format, generated on the GPU with the batch multiply."""
    from helpers import affine_to_wire
    from noble_curves_amd._native import POINT_BYTES
    order = Pt.Fn.ORDER
    ks = [(a + i * b) % order for i in range(n)]
    sc = torch.from_numpy(ints_to_le_bytes(ks).copy()).to(device)
    g = np.frombuffer(affine_to_wire(curve, Pt.BASE.toAffine()), dtype=np.uint8)
    base = torch.from_numpy(np.tile(g, (n, 1))).to(device)
    pb = POINT_BYTES[curve]
    out = torch.empty((n, pb), dtype=torch.uint8, device=device)
    inf = torch.empty((n,), dtype=torch.uint8, device=device)
    eng.mul_var_batch_dev(curve, n, dev_ptr(base), dev_ptr(sc), dev_ptr(out), dev_ptr(inf), stream)
    torch.cuda.synchronize()
    assert int(inf.sum().item()) == 0
    return out, ks


def gen_scalars(n, order_bits_safe, seed, device, edge_order=None):
    """uniform in [0, 2^order_bits_safe) (< group order), with k = 0, 1, n-1 at fixed indices."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    sc = torch.randint(0, 256, (n, 32), dtype=torch.uint8, generator=g)
    top_bits = order_bits_safe - 248
    sc[:, 31] &= (1 << top_bits) - 1
    if edge_order is not None and n >= 4:
        sc[0] = 0
        sc[1] = 0
        sc[1, 0] = 1
        sc[2] = torch.from_numpy(ints_to_le_bytes([edge_order - 1])[0].copy())
    return sc.to(device)


def scalars_to_ints(sc):
    arr = sc.cpu().numpy()
    return [int.from_bytes(arr[i].tobytes(), "little") for i in range(arr.shape[0])]


PREWARM_S = float(os.environ.get("NCG_BENCH_PREWARM_S", "0.05"))
PREWARM_DIST_STEPS = 6


class StepTimes(tuple):
    """(wall seconds, HIP-event milliseconds) of the K timed steps, plus the per-step distributions:
    `ev` = HIP-event time between consecutive steps on the launch stream, `wl` = host wall clock per step
    (for calls that synchronise inside - the MSM - this is the step's latency; for asynchronous launches it
    is the enqueue time and the event figure is the one that counts)."""

    def __new__(cls, wall, ev_ms, ev_steps, wl_steps):
        o = super().__new__(cls, (wall, ev_ms))
        o.ev_steps, o.wl_steps = ev_steps, wl_steps
        return o

    @staticmethod
    def _dist(v):
        s = sorted(v)
        return {"min": s[0], "median": s[len(s) // 2] if len(s) % 2 else 0.5 * (s[len(s) // 2 - 1] + s[len(s) // 2]),
                "max": s[-1], "list": [round(x, 4) for x in v]}

    def dist(self):
        return {"event_ms": self._dist(self.ev_steps), "wall_ms": self._dist(self.wl_steps)}


def time_steps(fn, steps, warmup, dist_on):
    """W warm-up steps, then exactly K steps bracketed by barrier + synchronize; returns a StepTimes:
    (wall seconds, HIP-event milliseconds on the launch stream) + per-step event / wall lists (an event is
    recorded between steps - no synchronisation is added inside the timed region)."""
    import torch.distributed as dist
    # clock pre-warm (untimed, before the W warm-up steps): the boost clock of an idle MI355X needs a few tens of
    # milliseconds of load to settle - the first workload after host-side setup otherwise reads ~5 % slow
    # (with several ranks the steps may be collectives, so the count must be the same on every rank: a fixed number)
    t_pre, cnt = time.perf_counter(), 0
    while (cnt < PREWARM_DIST_STEPS) if dist_on else (time.perf_counter() - t_pre < PREWARM_S and cnt < 64):
        fn()
        torch.cuda.synchronize()
        cnt += 1
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    ts = [0.0] * (steps + 1)
    t0 = time.perf_counter()
    ts[0] = t0
    evs[0].record()
    for i in range(steps):
        fn()
        evs[i + 1].record()
        ts[i + 1] = time.perf_counter()
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    return StepTimes(wall, evs[0].elapsed_time(evs[steps]),
                     [evs[i].elapsed_time(evs[i + 1]) for i in range(steps)],
                     [(ts[i + 1] - ts[i]) * 1e3 for i in range(steps)])


def time_pipelined(submit, collect, depth, steps, warmup, dist_on, check=None):
    """The K-step contract for MSMs IN FLIGHT: W warm-up steps, barrier + synchronize, then exactly K MSMs submitted
    over `depth` lanes (lane i % depth is collected - result on the host, host finish done - just before it is reused)
    and the last `depth` collected, synchronize + barrier.  Every collected result goes through `check`.
    Returns (wall seconds for the K MSMs, per-MSM completion intervals in ms)."""
    import torch.distributed as dist
    def run(k, timed):
        done = []
        t0 = time.perf_counter()
        for i in range(k):
            if i >= depth:
                r = collect(i % depth)
                done.append(time.perf_counter())
                if check:
                    check(r)
            submit(i % depth, i)
        for i in range(max(0, k - depth), k):                 # the MSMs still in flight, oldest first
            r = collect(i % depth)
            done.append(time.perf_counter())
            if check:
                check(r)
        return t0, done
    run(max(warmup, depth) + (PREWARM_DIST_STEPS if dist_on else 4), False)
    if not dist_on:
        # the same 50 ms of untimed load every other timed loop gets (time_steps: the boost clock settles over a few tens of
        # milliseconds after host-side set-up; a share of 0.5 ms measured over 40 jobs right after an idle gap read 0.60-0.63 ms,
        # the same jobs 0.49-0.50 ms once the clock had settled - tools/share_stream_test.py); ranks of a distributed run keep the
        # fixed step count above (the jobs may be collectives)
        t_pre = time.perf_counter()
        while time.perf_counter() - t_pre < PREWARM_S:
            run(2 * depth, False)
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    t0, done = run(steps, True)
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    marks = [t0] + done
    return wall, [round((marks[i + 1] - marks[i]) * 1e3, 4) for i in range(len(done))]


def hwmon_dir(dev_index):
    """sysfs hwmon directory of the GPU torch calls `dev_index` (the host's sysfs lists every card; this one by its PCI address)"""
    import glob
    try:
        pr = torch.cuda.get_device_properties(dev_index)
        bdf = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        hits = sorted(glob.glob("/sys/bus/pci/devices/%s/hwmon/hwmon*" % bdf))
        if hits:
            return hits[0]
    except Exception:  # noqa: BLE001 - a measurement beside the result, never a failure of the bench
        pass
    try:   # older torch without the PCI fields: the one GPU rocm-smi lists
        import re
        import subprocess
        txt = subprocess.run(["rocm-smi", "--showbus"], capture_output=True, text=True, timeout=20).stdout
        m = re.search(r"([0-9a-fA-F]{4}:[0-9a-fA-F]{2}:[0-9a-fA-F]{2}\.[0-9])", txt)
        hits = sorted(glob.glob("/sys/bus/pci/devices/%s/hwmon/hwmon*" % m.group(1).lower())) if m and dev_index == 0 else []
        return hits[0] if hits else None
    except Exception:  # noqa: BLE001
        return None


def power_state(fn, dev_index, seconds=1.2):
    """The shader clock and package power the box holds UNDER the headline workload (untimed, after the K timed steps): `fn` runs
    back to back for `seconds`, four steps per synchronisation, hwmon freq1_input / power1_input read in between.  The ladder is
    power-limited on every box of the pool: ms_per_step x sclk is what the code is responsible for."""
    h = hwmon_dir(dev_index)
    if not h:
        return None

    def rd(name):
        try:
            with open(os.path.join(h, name)) as f:
                return float(f.read().strip())
        except Exception:  # noqa: BLE001
            return None
    f_hz, p_uw = [], []
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(4):
            fn()
        a, b = rd("freq1_input"), rd("power1_input")
        if b is None:
            b = rd("power1_average")
        torch.cuda.synchronize()
        if time.perf_counter() - t0 > 0.3:          # the readings of the first 0.3 s still see the idle gap before
            if a:
                f_hz.append(a / 1e6)
            if b:
                p_uw.append(b / 1e6)
    if not f_hz:
        return None
    f_hz.sort()
    p_uw.sort()
    cap = rd("power1_cap")
    return {"sclk_mhz": [f_hz[0], f_hz[len(f_hz) // 2], f_hz[-1]], "power_w": [p_uw[0], p_uw[len(p_uw) // 2], p_uw[-1]] if p_uw else None,
            "power_cap_w": cap / 1e6 if cap else None, "samples": len(f_hz), "seconds": seconds,
            "source": "hwmon freq1_input / power1_input of the visible GPU, [min, median, max] over the samples after 0.3 s of back-to-back steps (untimed leg)"}


def max_over_ranks(x, dist_on, device):
    if not dist_on:
        return x
    import torch.distributed as dist
    t = torch.tensor([x], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks_bigint(v, order, dist_on, device):
    """(sum over ranks of a big integer) mod order, moved as 62-bit chunks."""
    if not dist_on:
        return v % order
    import torch.distributed as dist
    t = torch.tensor([v >> (62 * j) & ((1 << 62) - 1) for j in range(5)], dtype=torch.int64,
                     device=device if dist.get_backend() == "nccl" else "cpu")
    parts = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, t)
    return sum(sum(int(x) << (62 * j) for j, x in enumerate(p.tolist())) for p in parts) % order


def cpu_info():
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    node = None
    try:
        node = subprocess.run(["node", "--version"], capture_output=True, text=True, timeout=10).stdout.strip() or None
    except (OSError, subprocess.SubprocessError):
        pass
    from oracle import refjs
    ok = refjs.available()
    return {"cpu_model": model, "logical_cores": os.cpu_count(), "node_version": node,
            "reference_runnable": ok,
            "reference_note": ("the reference's own TypeScript sources run on this host: type-stripped by oracle/ref_js/downlevel.py into oracle/_ref/refjs.bundle "
                               "(types, generics, casts removed; every line of arithmetic untouched), @noble/hashes replaced by node:crypto, a few "
                               "library polyfills for Node 12 (kind: reference; the C port's figures are kept under cpu_baseline.port)") if ok else
                              ("the TypeScript reference needs Node >= 20.19 with type stripping and @noble/hashes (SURVEY 8c) and oracle/_ref/refjs.bundle is not built; "
                               "the CPU figures are the oracle's C restatement of the same algorithm (kind: port)")}


def cpu_baseline_rates(work, total_items, chunk, seconds, check=None):
    """Times `work(lo, hi)` (a ctypes call into oracle/c that releases the GIL) on one thread and on all
    hardware threads, each for about `seconds`; returns (rate_1, done_1, rate_all, done_all, threads)."""
    from concurrent.futures import ThreadPoolExecutor
    done, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds and done + chunk <= total_items:
        r = work(done, done + chunk)
        if check:
            check(done, done + chunk, r)
        done += chunk
    r1, d1 = done / (time.perf_counter() - t0), done
    threads = os.cpu_count() or 1
    if threads == 1 or done + chunk * threads > total_items:
        return r1, d1, None, 0, threads
    start = done
    per_thread = max(1, int(r1 * seconds / chunk))              # chunks each thread should take
    per_thread = min(per_thread, (total_items - start) // (chunk * threads))
    if per_thread < 1:
        return r1, d1, None, 0, threads

    span = per_thread * chunk                                   # ONE call per thread: the whole slice runs outside the GIL
    results = [None] * threads

    def run(t):
        lo = start + t * span
        results[t] = work(lo, lo + span)
        return span
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=threads) as ex:
        dall = sum(ex.map(run, range(threads)))
    dt = time.perf_counter() - t0
    if check:
        for t in range(threads):
            check(start + t * span, start + (t + 1) * span, results[t])
    return r1, d1, dall / dt, dall, threads


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: re-run this command line as N ranks (one per GPU) under
    torch.distributed.run on this node; the torchrun form the driver uses keeps working (WORLD_SIZE is then set)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log2n", type=int, default=20, help="items per GPU (2^log2n)")
    ap.add_argument("--workload", default="all", choices=["all", "secp256k1", "msm_g1", "msm_g2", "ed25519", "ntt"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=6.0, help="per CPU-baseline leg (1 thread, then all threads)")
    ap.add_argument("--no-live-pmc", action="store_true", help="do not run the rocprofv3 counter passes in this run")
    ap.add_argument("--quick-verify", action="store_true", help="(PMC child runs) skip the slow host-side cross-checks")
    ap.add_argument("--out", default=None, help="where the FULL result object goes (default gpurun_out/bench_full.json); stdout gets the compact line")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for dry runs)")
    ap.add_argument("--dist-dry-run", action="store_true",
                    help="run the multi-rank code path (nccl process group, the engine's RCCL communicator, sharded / in-flight MSM legs) with a world of ONE "
                         "rank: what a 1-GPU box can execute of `--gpus N`; the line is labelled dist_dry_run")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))
    # stdout carries exactly ONE line - the result - and it must be the LAST thing on it: libraries write there too (RCCL's
    # version banner sits in the C stdio buffer until the process exits and lands AFTER the JSON line; gloo announces its peers).
    # File descriptor 1 is pointed at stderr for everything else; the line goes out through a private duplicate of the real one.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    sys.stdout = sys.stderr
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist_on = world > 1 or args.dist_dry_run
    if args.dist_dry_run and world == 1:
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("MASTER_PORT", "29531")
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback exists)"
    if dist_on and args.backend == "nccl" and torch.cuda.device_count() < world and not os.environ.get("NCG_BENCH_FORCE_NCCL"):
        # fewer GPUs than ranks (a 1-GPU box): RCCL refuses two ranks on one device, so the ranks share GPUs and
        # exchange through gloo - the native per-shard phase and the native combine still run (host-staged slots)
        args.backend = "gloo"
        # several processes on ONE GPU: each would add its own top-priority hardware queues for the asynchronous MSM lanes (comm.hip
        # lane_init) - eight ranks x three lanes oversubscribe the queues the command processors keep resident and the in-flight legs
        # crawl (18 -> 50 ms per MSM in the 8-rank self-launch).  Plain lane streams there: the library's knob for shared GPUs.
        os.environ.setdefault("NCG_LANE_QUEUES", "0")
    dev_index = local_rank % torch.cuda.device_count()   # (dry runs may put several ranks on one GPU)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(args.backend)
    assert world == args.gpus, "WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus)

    from noble_curves_amd import get_engine
    from noble_curves_amd._native import BLS12_381_G1, BLS12_381_G2, ED25519, SECP256K1
    from noble_curves_amd.distributed import init_comm, msm_sharded
    from helpers import wire_to_affine
    from oracle import cport
    from oracle.curves import BLS_R, BlsG1, BlsG2, SECP256K1_N, Secp256k1, makeRng

    eng = get_engine(dev_index)
    # a real (non-null) stream: kernels, copies and the timing events all go on it
    tstream = torch.cuda.Stream(device=device)
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream
    assert stream != 0
    n = 1 << args.log2n
    K, W = args.steps, args.warmup
    cpu_leg = rank == 0 and world == 1 and not args.no_cpu_baseline
    live = None
    if rank == 0 and world == 1 and not args.no_live_pmc and not args.quick_verify:
        live = pmc_live(args.workload, args.log2n)
    pmc = Pmc(live)
    native_multi = init_comm(eng, device, single_ok=args.dist_dry_run) if dist_on else False
    result = {}
    extra = {}
    host = cpu_info() if rank == 0 else {}

    from oracle import refjs
    ref_ok = rank == 0 and refjs.available()

    def with_reference(port_entry, ref_info, unit, sample):
        """cpu_baseline = the REFERENCE's own TypeScript code on this host's Node (oracle/_ref/refjs.bundle: /root/reference/src type-stripped
        by oracle/ref_js/downlevel.py; one thread - the reference is single-threaded JS) when that build is present, with the C
        port's figures (1 thread and all threads) kept beside it; otherwise the port alone."""
        if not ref_info:
            return port_entry
        e = {"value": ref_info["per_s"], "unit": unit, "cores": 1, "kind": "reference", "sample": sample,
             "runtime": "node %s (BigInt), @noble/hashes replaced by node:crypto" % ref_info.get("node"),
             "cpu_model": host.get("cpu_model"), "logical_cores": host.get("logical_cores"), "port": port_entry}
        return e

    def baseline_entry(r1, d1, rall, dall, threads, unit, sample):
        e = {"value": r1, "unit": unit, "cores": 1, "kind": "port", "sample": sample % d1,
             "cpu_model": host.get("cpu_model"), "logical_cores": host.get("logical_cores")}
        if rall:
            e["all_threads"] = {"value": rall, "cores": threads, "items": dall}
        return e

    # ------------------------------------------------------------------ secp256k1 batch multiply
    if args.workload in ("all", "secp256k1"):
        rng = makeRng(0x6E6F626C6502 + rank)
        a, b = rng.rndBelow(SECP256K1_N - 1) + 1, rng.rndBelow(SECP256K1_N - 1) + 1
        pts, pks = gen_points(eng, SECP256K1, Secp256k1, n, a, b, device, stream)
        sc = gen_scalars(n, 255, 1234 + rank, device, edge_order=SECP256K1_N)
        out = torch.empty((n, 64), dtype=torch.uint8, device=device)
        inf = torch.empty((n,), dtype=torch.uint8, device=device)

        def step():
            eng.mul_var_batch_dev(SECP256K1, n, dev_ptr(pts), dev_ptr(sc), dev_ptr(out), dev_ptr(inf), stream)

        st_secp = time_steps(step, K, W, dist_on)
        wall, ev_ms = st_secp
        pstate = power_state(step, dev_index) if (rank == 0 and world == 1 and not args.quick_verify) else None
        wall = max_over_ranks(wall, dist_on, device)
        ev_ms = max_over_ranks(ev_ms, dist_on, device)
        # ---- verification (outside the timed region)
        ks = scalars_to_ints(sc)
        expect_sum = sum(k * p for k, p in zip(ks, pks)) % SECP256K1_N
        ones = torch.zeros((n, 32), dtype=torch.uint8, device=device)
        ones[:, 0] = 1
        tot, tot_inf = eng.msm_dev(SECP256K1, n, dev_ptr(out), dev_ptr(ones), stream)
        assert wire_to_affine(SECP256K1, tot) == Secp256k1.BASE.multiplyUnsafe(expect_sum).toAffine(), \
            "full-size checksum mismatch"
        assert int(inf.sum().item()) == 1 and int(inf[0].item()) == 1   # only k = 0 gives infinity
        S = 512
        o_c, _ = cport.multiply_unsafe("secp256k1", pts[:S].cpu().numpy(), sc[:S].cpu().numpy())
        assert np.array_equal(o_c, out[:S].cpu().numpy()), "sample mismatch vs oracle"
        kern_ms = ev_ms / K
        rate = world * n * K / wall
        alg_bytes = 160.0 * n          # 64 B point + 32 B scalar in, 64 B out
        traffic, tsrc = pmc.traffic("secp256k1") if args.log2n == 20 else (None, None)
        result = {
            "metric": "secp256k1_scalar_mults_per_sec", "value": rate, "unit": "scalar-mults/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": wall / K * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32",
            "data": "synthetic: P_i=(a+i*b)G, k_i uniform in [0,2^255) with k=0,1,n-1 planted; seed xorshift64",
            "config": {"workload": "secp256k1 batch variable-base multiplyUnsafe (GLV), 2^%d pairs per GPU"
                       % args.log2n, "items_per_gpu": n, "parallelism": "shard-by-index x%d" % world},
            "step_times": st_secp.dist(),
            "roofline": {"bound": "valu", "bound_note": BOUND_NOTE, "achieved": alg_bytes / (kern_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": alg_bytes / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": tsrc,
                         "kernel": "k_mul_var_gtab<CurveSecpI,4,3> (+ k_jac_batch_affine<16>, included in kernel_ms); "
                                   "traffic = the dominant kernel's, incl. its per-item window table in device memory",
                         "kernel_ms": kern_ms,
                         "valu": valu_block(pmc, "secp256k1", kern_ms * 1e-3, 3.4e5 * n, secp_mads_per_mult() * n)},
        }
        if pstate:
            pstate["ms_per_step_x_sclk"] = wall / K * 1e3 * pstate["sclk_mhz"][1]
            result["power_state"] = pstate
        if not dist_on and not args.quick_verify:   # (the PMC / kernel-stats child runs keep to the headline launches)
            # end to end through the host-pointer entry point (ncg_mul_var_batch): pinned-once host buffers, chunked H2D / kernels / D2H
            e_p, e_s = pts.cpu().numpy(), sc.cpu().numpy()
            e_o, e_i = np.zeros((n, 64), np.uint8), np.zeros((n,), np.uint8)
            for arr in (e_p, e_s, e_o, e_i):
                eng.host_register(arr)
            try:
                def estep():
                    eng._check(eng.lib.ncg_mul_var_batch(eng.h, SECP256K1, n, e_p.ctypes.data, e_s.ctypes.data, e_o.ctypes.data, e_i.ctypes.data))
                st_e = time_steps(estep, K, W, False)
                assert np.array_equal(e_o, out.cpu().numpy()), "host-pointer batch multiply differs from the device-buffer one"
                result["end_to_end"] = {"value": n * K / st_e[0], "unit": "scalar-mults/s", "ms_per_batch": st_e[0] / K * 1e3,
                                        "bytes_in_out": int(e_p.nbytes + e_s.nbytes + e_o.nbytes + e_i.nbytes),
                                        "note": "ncg_mul_var_batch on host buffers pinned once (96 MB in, 65 MB out per 2^20 pairs), chunks pipelined over "
                                                "three streams; PCIe-inclusive, never the headline value; output compared bit-exactly"}
            finally:
                for arr in (e_p, e_s, e_o, e_i):
                    eng.host_unregister(arr)
        if cpu_leg:
            pts_h, sc_h, out_h = pts.cpu().numpy(), sc.cpu().numpy(), out.cpu().numpy()

            def work(lo, hi):
                return cport.multiply_unsafe("secp256k1", pts_h[lo:hi], sc_h[lo:hi])[0]

            def check(lo, hi, r):
                assert np.array_equal(r, out_h[lo:hi]), "CPU baseline output differs from the GPU's"
            # BASELINE configs[0] (benchmark/point.ts:20-32, the reference's own CPU case): ONE point, Point.multiply and
            # Point.multiplyUnsafe on 1 000 random scalars and on the benchmark's literal 2^180 - 15820, through the port
            # (Point.multiply = the blinded constant-time fixed-window path an un-precomputed point takes, curve.ts:663-729)
            rng0 = makeRng(0x6E6F626C6501)
            k0 = [rng0.rndBelow(SECP256K1_N - 1) + 1 for _ in range(1000)] + [(1 << 180) - 15820]
            p0 = np.tile(pts_h[7], (len(k0), 1))
            k0w = ints_to_le_bytes(k0)
            bl0 = np.frombuffer(bytes((rng0.rnd64() >> 11) & 0xFF for _ in range(16 * len(k0))), np.uint8).reshape(-1, 16)
            t0 = time.perf_counter()
            o_ct, _ = cport.multiply(p0, k0w, bl0)
            t_ct = time.perf_counter() - t0
            t0 = time.perf_counter()
            o_un, _ = cport.multiply_unsafe("secp256k1", p0, k0w)
            t_un = time.perf_counter() - t0
            g0 = torch.empty((len(k0), 64), dtype=torch.uint8, device=device)
            i0 = torch.empty((len(k0),), dtype=torch.uint8, device=device)
            d_p0, d_k0 = torch.from_numpy(p0.copy()).to(device), torch.from_numpy(k0w.copy()).to(device)
            torch.cuda.synchronize()
            eng.mul_var_batch_dev(SECP256K1, len(k0), dev_ptr(d_p0), dev_ptr(d_k0), dev_ptr(g0), dev_ptr(i0), stream)
            torch.cuda.synchronize()
            assert np.array_equal(o_ct, o_un), "configs[0]: Point.multiply and Point.multiplyUnsafe ports disagree"
            assert np.array_equal(o_un, g0.cpu().numpy()), "configs[0]: the GPU batch multiply differs from the CPU port"
            extra["configs0_point_multiply"] = {
                "metric": "secp256k1_point_multiply_ops_per_sec", "unit": "ops/s", "kind": "port", "cores": 1,
                "Point_multiply": {"value": len(k0) / t_ct, "us_per_op": t_ct / len(k0) * 1e6,
                                   "shape": "fixedWindowCT W=5, 128-bit blind: 77 windows x (5 dbl + 1 add) + 31-add table"},
                "Point_multiplyUnsafe": {"value": len(k0) / t_un, "us_per_op": t_un / len(k0) * 1e6,
                                         "shape": "GLV split + joint wNAF-4 (curve.ts:820-836)"},
                "sample": "1 point, 1000 random scalars (xorshift64 seed 0x6e6f626c6501) + the literal 2^180-15820 of benchmark/point.ts:20; "
                          "oracle/c restatement (RCB formulas, 64-bit Montgomery limbs), results equal and equal to the GPU batch multiply",
                "cpu_model": host.get("cpu_model"),
                "note": "BASELINE configs[0] is the reference's CPU plumbing case; `reference` (when present) = the reference's own code on this host, the figures above = the C port"}
            r1, d1, rall, dall, thr = cpu_baseline_rates(work, n, 500, args.cpu_seconds, check)
            port_e = baseline_entry(r1, d1, rall, dall, thr, "scalar-mults/s",
                                    "first %d pairs of the same batch through oracle/c (RCB + GLV wNAF-4), outputs "
                                    "compared bit-exactly with the GPU's; all_threads: the next pairs, one chunk stream per thread")
            ref_info = None
            if ref_ok:
                M = 4096
                o_r, ref_info = refjs.multiply(SECP256K1, pts_h[:M], sc_h[:M], unsafe=True)
                assert np.array_equal(o_r, out_h[:M]), "the reference's Point.multiplyUnsafe differs from the GPU batch multiply"
                pbj = refjs.point_bench(1.5)
                assert pbj["equal"]
                extra["configs0_point_multiply"]["reference"] = {
                    "kind": "reference", "cores": 1, "unit": "ops/s", "Point_mul": pbj["Point_mul"], "Point_mulUns": pbj["Point_mulUns"],
                    "Point_mul_random_scalars": pbj["Point_mul_random"], "Point_mulUns_random_scalars": pbj["Point_mulUns_random"],
                    "note": "benchmark/point.ts:20-32 run by the reference's own code on this host (node %s): Point.multiply / multiplyUnsafe of one "
                            "public-key point by 2^180 - 15820 (BASE precomputed with W = 6 as the benchmark does), and by 1 000 random scalars" % pbj["node"]}
            result["cpu_baseline"] = with_reference(port_e, ref_info, "scalar-mults/s",
                                                    "first 4096 pairs of the same batch through the reference's Point.multiplyUnsafe (weierstrass.ts:915-928), "
                                                    "outputs compared bit-exactly with the GPU's")

    # ------------------------------------------------------------------ bls12-381 G1 / G2 MSM
    def msm_workload(curve, Pt, cname, nn, seed, alg_b, ref_mac_pt, key):
        rng = makeRng(seed + rank)
        a, b = rng.rndBelow(BLS_R - 1) + 1, rng.rndBelow(BLS_R - 1) + 1
        pts, pks = gen_points(eng, curve, Pt, nn, a, b, device, stream)
        sc = gen_scalars(nn, 254, 777 + seed % 1000 + rank, device)
        sc[::17] = 0                                         # test/slow-curves.test.ts:215
        ks = scalars_to_ints(sc)
        local_expect = sum(k * p for k, p in zip(ks, pks)) % BLS_R
        holder = {}

        def step():
            holder["r"] = msm_sharded(eng, curve, nn, dev_ptr(pts), dev_ptr(sc), stream, device, n_max=nn)

        st_msm = time_steps(step, K, W, dist_on)
        wall, ev_ms = st_msm
        pstate_m = power_state(step, dev_index, 0.9) if (rank == 0 and world == 1 and not args.quick_verify) else None
        wall = max_over_ranks(wall, dist_on, device)
        expect = sum_over_ranks_bigint(local_expect, BLS_R, dist_on, device)
        got, got_inf = holder["r"]
        assert wire_to_affine(curve, got) == Pt.BASE.multiplyUnsafe(expect).toAffine(), "MSM mismatch"
        plan = eng.msm_plan_info(curve, nn)
        c, nwin = plan["c"], plan["nwin"]
        traffic, tsrc = pmc.traffic(key) if (curve == BLS12_381_G1 and args.log2n == 20) or (curve == BLS12_381_G2 and args.log2n == 20) else (None, None)
        entry = {"metric": "bls12_381_%s_msm_points_per_sec" % cname, "value": world * nn * K / wall, "unit": "points/s",
                 "ms_per_msm": wall / K * 1e3, "step_times": st_msm.dist(),
                 "points_per_gpu": nn, "total_points": world * nn, "scaling": "weak", "window_plan": plan,
                 "multi_gpu": ("ncg_msm_sharded_dev: RCCL all-gather of grouped window sums + on-device add" if native_multi
                               else (("%s: ncg_msm_shard_local_dev -> torch.distributed all-gather of the slots (window-plan header + "
                                      "grouped window sums) -> ncg_msm_shard_combine (native header check, adding kernel, finish)"
                                      % ("host-staged exchange over gloo (ranks share GPUs)" if args.backend == "gloo" else "host-staged exchange (native RCCL communicator unavailable)"))
                                     if dist_on else "single GPU")),
                 "roofline": {"bound": "valu", "bound_note": BOUND_NOTE, "achieved": alg_b * nn / (wall / K) / 1e9, "peak": HBM_PEAK_GBS,
                              "unit": "GB/s", "frac": alg_b * nn / (wall / K) / 1e9 / HBM_PEAK_GBS,
                              "traffic": traffic, "traffic_source": tsrc,
                              "kernel": "k_msm_accum (dominant; achieved is for the whole MSM incl. the host finish, traffic for that kernel)",
                              "valu": valu_block(pmc, key, wall / K, ref_mac_pt * nn,
                                                 (g1_msm_mads_per_point(nwin, nn, 1 << (c - 1), True) * (3 if curve == BLS12_381_G2 else 1)) * nn)}}
        if pstate_m:
            pstate_m["ms_per_msm_x_sclk"] = wall / K * 1e3 * pstate_m["sclk_mhz"][1]
            entry["power_state"] = pstate_m
        if not dist_on and not args.quick_verify:   # (the PMC / kernel-stats child runs keep to the headline launches)
            # end to end through the HOST-pointer entry point (the boundary the N-API addon binds: ncg_msm): inputs in pinned host
            # memory (ncg_host_register once), scalars first, the points in parts accumulated while the next part crosses PCIe
            pts_h, sc_h = pts.cpu().numpy(), sc.cpu().numpy()
            eng.host_register(pts_h)
            eng.host_register(sc_h)
            try:
                eh = {}

                def estep():
                    eh["r"] = eng.msm(curve, pts_h, sc_h)

                st_e = time_steps(estep, K, W, False)
                assert np.array_equal(eh["r"][0], got), "host-pointer MSM differs from the device-buffer MSM"
                entry["end_to_end"] = {"value": nn * K / st_e[0], "unit": "points/s", "ms_per_msm": st_e[0] / K * 1e3,
                                       "bytes_in": int(pts_h.nbytes + sc_h.nbytes), "step_times": {"wall_ms": st_e.dist()["wall_ms"]},
                                       "note": "ncg_msm on host buffers pinned once: H2D of every byte + MSM + result, per call (PCIe-inclusive; "
                                               "never the headline value); result compared bit-exactly with the device-buffer MSM"}
            finally:
                eng.host_unregister(pts_h)
                eng.host_unregister(sc_h)
            # several MSMs in flight (ncg_msm_async_submit / _collect): the dependent tail of one MSM (narrow fold levels,
            # per-window tail, D2H, host Horner) overlaps the sort / accumulate kernels of the next
            def chk(r):
                assert np.array_equal(r[0], got), "pipelined MSM differs from the synchronous one"
            pipe = {}
            for depth in (2, 3):
                pw, iv = time_pipelined(lambda lane, i: eng.msm_async_submit(lane, curve, nn, dev_ptr(pts), dev_ptr(sc), stream),
                                        lambda lane: eng.msm_async_collect(lane, curve), depth, K, W, False, chk)
                pipe["depth%d" % depth] = {"value": nn * K / pw, "unit": "points/s", "ms_per_msm": pw / K * 1e3, "completion_intervals_ms": iv}
            entry["pipelined"] = dict(pipe, note="K MSMs submitted over `depth` lanes (own stream / workspace / pinned landing area each), lane "
                                                 "i % depth collected before reuse; every result compared with the synchronous MSM; same K-step "
                                                 "contract (barrier + synchronize around the K MSMs)")
            # ONE rank's share of this MSM in the window-sharded multi-GPU mode, emulated on this GPU:
            # part r of G runs windows [w0, w0 + cnt) of ALL points; `latency` = part 0 synchronously + the combine / finish
            # every rank runs on the G gathered slots; `pipelined` = the parts of successive MSMs in flight on the lanes
            # (average over all G parts).  The G slots are combined and checked, so the timed path is the shipped one.
            # Not included: the all-gather of G x 29 KB (58 KB on G2) over xGMI.
            shares = {}
            wres = eng.upload_points(curve, pts.cpu().numpy())          # the mode's premise: the set is resident on every GPU
            for G in (2, 4, 8):
                slots = [eng.msm_shard_windows_local_dev(curve, nn, r, G, 0, dev_ptr(sc), stream, wres) for r in range(G)]
                stack = np.stack(slots)
                cg, _ = eng.msm_shard_combine(curve, nn, stack, stream)
                assert np.array_equal(cg, got), "window-sharded MSM (%d parts) differs from the single-GPU MSM" % G
                st_l = time_steps(lambda: eng.msm_shard_windows_local_dev(curve, nn, 0, G, 0, dev_ptr(sc), stream, wres), K, W, False)
                st_c = time_steps(lambda: eng.msm_shard_combine(curve, nn, stack, stream), K, W, False)
                coll = {}

                def sub(lane, i, G=G):
                    eng.msm_async_submit(lane, curve, nn, 0, dev_ptr(sc), stream, wres, eng.async_part(i % G, G))
                    coll[lane] = i % G

                def col(lane):
                    return coll[lane], eng.msm_async_collect_slot(lane, curve)
                seen = {}
                jobs = G * max(5, (2 * K + G - 1) // G)          # whole MSMs' worth of parts: 40 at the driver's K = 20
                pw, _ = time_pipelined(sub, col, 3, jobs, W, False, lambda r: seen.__setitem__(r[0], r[1]))
                cg2, _ = eng.msm_shard_combine(curve, nn, np.stack([seen[r] for r in range(G)]), stream)
                assert np.array_equal(cg2, got), "window-sharded MSM (%d parts, pipelined) differs from the single-GPU MSM" % G
                lat = (st_l[0] + st_c[0]) / K * 1e3
                shares["G%d" % G] = {"latency_ms": lat, "local_part0_ms": st_l[0] / K * 1e3, "combine_finish_ms": st_c[0] / K * 1e3,
                                     "pipelined_part_ms": pw / jobs * 1e3, "speedup_latency": (wall / K * 1e3) / lat,
                                     "speedup_pipelined": (wall / K * 1e3) / (pw / jobs * 1e3)}
            wres.free()
            entry["window_share"] = dict(shares, note="per-rank share of ONE %d-point MSM cut by windows over G ranks (resident set, stored form), emulated on one GPU; speedup_* = "
                                                      "this run's single-GPU ms_per_msm / share; the xGMI all-gather (~30 us) is not included" % nn)
        if not dist_on:
            # the same MSM on a resident set verified to lie in the prime-order subgroup (ncg_points_verify_subgroup,
            # once per set): the scalars are split along the curve endomorphism (csrc/endo.hpp) - same group
            # element, half (G1) / a quarter (G2) of the windows.  pippenger itself accepts arbitrary curve
            # points, so the headline above never assumes this.
            res = eng.upload_points(curve, pts.cpu().numpy())
            t0 = time.perf_counter()
            bad = res.verify_subgroup()
            verify_ms = (time.perf_counter() - t0) * 1e3
            assert bad == -1 and res.in_subgroup
            rh = {}

            def rstep():
                rh["r"] = res.msm_dev(dev_ptr(sc), stream)

            st_res = time_steps(rstep, K, W, False)
            rwall, _ = st_res
            assert np.array_equal(rh["r"][0], got), "endomorphism MSM differs from the generic MSM"
            entry["resident_subgroup_set"] = {
                "value": nn * K / rwall, "unit": "points/s", "ms_per_msm": rwall / K * 1e3, "step_times": st_res.dist(),
                "verify_once_ms": verify_ms,
                "note": "ncg_msm_resident_dev on a set that passed the reference's isTorsionFree test on every point "
                        "(bls12-381.ts:567-577 / :599-601) at upload; result compared bit-exactly with the generic MSM above"}
            # the same set with interleavedMSMUnsafe's precomputation in device form (ncg_points_precompute: window-shifted
            # copies, one shared bucket set; curve.ts:907-959 builds its per-point tables once too)
            t0 = time.perf_counter()
            pre_ok = res.precompute()
            torch.cuda.synchronize()
            pre_ms = (time.perf_counter() - t0) * 1e3
            if pre_ok:
                ph = {}

                def pstep():
                    ph["r"] = res.msm_dev(dev_ptr(sc), stream)

                st_pre = time_steps(pstep, K, W, False)
                assert np.array_equal(ph["r"][0], got), "precomputed-set MSM differs from the generic MSM"
                entry["resident_subgroup_set"]["precomputed"] = {
                    "value": nn * K / st_pre[0], "unit": "points/s", "ms_per_msm": st_pre[0] / K * 1e3, "step_times": st_pre.dist(),
                    "precompute_once_ms": pre_ms,
                    "note": "ncg_points_precompute on the verified set: shifted copies 2^(16 w) of every endomorphism image, all "
                            "windows add into one bucket set (one fold, no Horner across windows); result compared bit-exactly"}
            res.free()
        sub = {"pts": pts, "sc": sc, "ks": ks, "pks": pks}
        return entry, sub

    def strong_by_windows(curve, Pt, nn, seed):
        """configs[3] / [4] as written - ONE nn-point MSM on all GPUs - in the window-sharded mode: every rank holds the same
        nn points (a replicated / resident set: the fixed bases of a prover) and all nn scalars, rank r runs its range of the
        windows (digits, sort, accumulate and the throughput part of the fold divide by the rank count), ONE all-gather of
        the grouped window sums, concatenation, Horner on every rank.  Timed synchronously (latency of one MSM) and with 3
        MSMs in flight per rank (sustained rate)."""
        from noble_curves_amd.distributed import msm_sharded_windows
        rng = makeRng(seed)                                   # the SAME set on every rank
        a, b = rng.rndBelow(BLS_R - 1) + 1, rng.rndBelow(BLS_R - 1) + 1
        pts, pks = gen_points(eng, curve, Pt, nn, a, b, device, stream)
        sc = gen_scalars(nn, 254, 4242 + seed % 1000, device)
        sc[::17] = 0
        ks = scalars_to_ints(sc)
        exp = Pt.BASE.multiplyUnsafe(sum(k * p for k, p in zip(ks, pks)) % BLS_R).toAffine()
        res = eng.upload_points(curve, pts.cpu().numpy())     # resident on every GPU (stored form built once)
        hs = {}

        # the N = 1 figure OF THIS RUN: the same MSM on the same resident set by one GPU alone (every rank runs it on its own GPU at
        # the same time - no collective; the slowest rank's time is kept), so that speedup_vs_n1 compares like with like
        def step_1():
            hs["one"] = res.msm_dev(dev_ptr(sc), stream)

        st_1 = time_steps(step_1, K, W, dist_on)
        wall_1 = max_over_ranks(st_1[0], dist_on, device)
        assert wire_to_affine(curve, hs["one"][0]) == exp, "resident-set MSM mismatch"

        def step_w():
            hs["r"] = msm_sharded_windows(eng, curve, nn, 0, dev_ptr(sc), stream, device, res)

        st_w = time_steps(step_w, K, W, dist_on)
        wall_w = max_over_ranks(st_w[0], dist_on, device)
        assert wire_to_affine(curve, hs["r"][0]) == exp, "window-sharded MSM mismatch"
        out = {"metric": "bls12_381_%s_msm_points_per_sec" % ("g1" if curve == BLS12_381_G1 else "g2"), "value": nn * K / wall_w, "unit": "points/s",
               "ms_per_msm": wall_w / K * 1e3, "total_points": nn, "scaling": "strong", "mode": "windows", "step_times": st_w.dist(),
               "window_plan": eng.msm_plan_info(curve, nn),
               "transport": "ncg_msm_sharded_windows_dev (RCCL all-gather inside the C ABI)" if native_multi else "host-staged slots over torch.distributed (%s)" % args.backend,
               "note": "every rank holds all %d points (resident set) and scalars; rank r runs windows [w0, w0 + cnt); slots concatenated" % nn}
        out["ms_per_msm_n1"] = wall_1 / K * 1e3
        out["speedup_vs_n1"] = wall_1 / wall_w
        out["n1_note"] = "ms_per_msm_n1 = the same MSM on the same resident set by ONE GPU (ncg_msm_resident_dev), measured in this run on every rank at once, slowest rank"
        nwin_plan = out["window_plan"]["nwin"]
        out["world"] = world
        # csrc/msm_shard.hpp msm_shard_window_range: contiguous ranges, sizes differ by at most one
        out["windows_per_rank"] = [nwin_plan // world + (1 if r < nwin_plan % world else 0) for r in range(world)]
        if native_multi:
            out["rccl_ranks"] = eng.comm_count()[0]           # ncclCommCount of the communicator the all-gather ran on

        def chk(r):
            assert wire_to_affine(curve, r[0]) == exp, "pipelined window-sharded MSM mismatch"
        if not native_multi and world > 1:
            # the host-staged exchange with parts in flight (ranks sharing GPUs, transports other than RCCL): this rank's part of
            # MSM i on lane i % 3 (NCG_MSM_ASYNC_PART), its slot collected just before the lane is reused, the slots all-gathered
            # by torch.distributed, ncg_msm_shard_combine on every rank
            from noble_curves_amd.distributed import all_gather_slots

            def collect_staged(lane):
                slot = eng.msm_async_collect_slot(lane, curve)
                return eng.msm_shard_combine(curve, nn, all_gather_slots(slot, device), stream)
            pwall, iv = time_pipelined(lambda lane, i: eng.msm_async_submit(lane, curve, nn, 0, dev_ptr(sc), stream, res, eng.async_part(rank, world)),
                                       collect_staged, 3, K, W, dist_on, chk)
            pwall = max_over_ranks(pwall, dist_on, device)
            out["pipelined"] = {"value": nn * K / pwall, "unit": "points/s", "ms_per_msm": pwall / K * 1e3, "depth": 3, "completion_intervals_ms": iv,
                                "speedup_vs_n1": wall_1 / pwall,
                                "note": "3 parts in flight per rank (ncg_msm_async_submit with NCG_MSM_ASYNC_PART(rank, world) -> slot -> all-gather over "
                                        "torch.distributed -> ncg_msm_shard_combine); every result checked"}
        if native_multi:
            pwall, iv = time_pipelined(lambda lane, i: eng.msm_async_submit(lane, curve, nn, 0, dev_ptr(sc), stream, res, eng.ASYNC_WINDOWS),
                                       lambda lane: eng.msm_async_collect(lane, curve), 3, K, W, dist_on, chk)
            pwall = max_over_ranks(pwall, dist_on, device)
            out["pipelined"] = {"value": nn * K / pwall, "unit": "points/s", "ms_per_msm": pwall / K * 1e3, "depth": 3, "completion_intervals_ms": iv,
                                "speedup_vs_n1": wall_1 / pwall,
                                "note": "3 window-sharded MSMs in flight per rank (ncg_msm_async_submit with NCG_MSM_ASYNC_WINDOWS); every result checked"}
        res.free()
        return out

    if args.workload in ("all", "msm_g1"):
        msm, sub = msm_workload(BLS12_381_G1, BlsG1, "g1", n, 0x6D736D0000000003, 128.0, 9.45e4, "msm_g1")
        if cpu_leg:
            pts_h, sc_h = sub["pts"].cpu().numpy(), sub["sc"].cpu().numpy()
            m = min(n, 1 << 17)
            t0 = time.perf_counter()
            o_c, i_c = cport.pippenger("bls12_381_g1", pts_h[:m], sc_h[:m])
            dt = time.perf_counter() - t0
            g_s, _ = eng.msm_dev(BLS12_381_G1, m, dev_ptr(sub["pts"]), dev_ptr(sub["sc"]), stream)
            assert np.array_equal(o_c, g_s), "MSM sample mismatch vs oracle pippenger"
            thr = os.cpu_count() or 1
            rall = None
            if thr > 1 and n >= m * 2:
                from concurrent.futures import ThreadPoolExecutor
                mm = min(m, n // thr)
                t0 = time.perf_counter()
                with ThreadPoolExecutor(max_workers=thr) as ex:
                    list(ex.map(lambda t: cport.pippenger("bls12_381_g1", pts_h[t * mm:(t + 1) * mm], sc_h[t * mm:(t + 1) * mm]), range(thr)))
                rall = thr * mm / (time.perf_counter() - t0)
            port_e = baseline_entry(m / dt, m, rall, thr * (min(m, n // thr)) if rall else 0, thr, "points/s",
                                    "first %d points of the same MSM through oracle/c pippenger (curve.ts:863-905 "
                                    "restated), result compared bit-exactly with the GPU MSM on the same subset; "
                                    "all_threads: one independent MSM of n/threads points per thread (points/s summed)")
            ref_info = None
            if ref_ok:
                mr = min(n, 1 << 12)
                o_r, ref_info = refjs.pippenger(BLS12_381_G1, pts_h[:mr], sc_h[:mr])
                g_r, _ = eng.msm_dev(BLS12_381_G1, mr, dev_ptr(sub["pts"]), dev_ptr(sub["sc"]), stream)
                assert np.array_equal(o_r, g_r), "the reference's pippenger differs from the GPU MSM"
            msm["cpu_baseline"] = with_reference(port_e, ref_info, "points/s",
                                                 "first 4096 points of the same MSM through the reference's pippenger (abstract/curve.ts:863-905), one call; "
                                                 "result compared bit-exactly with the GPU MSM on the same subset")
        extra["msm_g1"] = msm
        if dist_on:
            # strong scaling (configs[3] as written): ONE 2^log2n-point MSM whose points are split
            # across the ranks (n/world each), combined inside the C ABI
            ns = n // world
            sub_expect = sum(k * p for k, p in zip(sub["ks"][:ns], sub["pks"][:ns])) % BLS_R
            hs = {}

            def step_strong():
                hs["r"] = msm_sharded(eng, BLS12_381_G1, ns, dev_ptr(sub["pts"]), dev_ptr(sub["sc"]), stream, device, n_max=ns)

            st_s = time_steps(step_strong, K, W, dist_on)
            wall_s, _ = st_s
            wall_s = max_over_ranks(wall_s, dist_on, device)
            tot = sum_over_ranks_bigint(sub_expect, BLS_R, dist_on, device)
            got_s, _ = hs["r"]
            assert wire_to_affine(BLS12_381_G1, got_s) == BlsG1.BASE.multiplyUnsafe(tot).toAffine(), "strong MSM mismatch"
            by_points = {"metric": "bls12_381_g1_msm_points_per_sec", "value": ns * world * K / wall_s,
                         "unit": "points/s", "ms_per_msm": wall_s / K * 1e3, "total_points": ns * world,
                         "points_per_gpu": ns, "scaling": "strong", "mode": "points", "step_times": st_s.dist()}
            extra["msm_g1_strong"] = strong_by_windows(BLS12_381_G1, BlsG1, n, 0x6D736D0000001003)
            extra["msm_g1_strong"]["by_points"] = by_points
        if not result:
            result = dict(msm)
            result.update({"n_gpus": world, "steps": K, "warmup": W, "ms_per_step": msm["ms_per_msm"],
                           "higher_is_better": True, "vs_baseline": None, "dtype": "u32",
                           "data": "synthetic: P_i=(a+i*b)G1, s_i uniform in [0,2^254), every 17th zero",
                           "config": {"workload": "bls12-381 G1 Pippenger MSM, 2^%d points per GPU" % args.log2n}})
            extra.pop("msm_g1", None)

    if args.workload in ("all", "msm_g2"):
        n2 = max(1, n >> 2)                                   # 2^18 at the default size (configs[4])
        msm2, sub2 = msm_workload(BLS12_381_G2, BlsG2, "g2", n2, 0x6D736D0000000004, 224.0, 3.0e5, "msm_g2")
        if cpu_leg:
            pts2_h, sc2_h = sub2["pts"].cpu().numpy(), sub2["sc"].cpu().numpy()
            m2 = min(n2, 1 << 15)
            t0 = time.perf_counter()
            o_c2, _ = cport.pippenger("bls12_381_g2", pts2_h[:m2], sc2_h[:m2])
            dt2 = time.perf_counter() - t0
            g_s2, _ = eng.msm_dev(BLS12_381_G2, m2, dev_ptr(sub2["pts"]), dev_ptr(sub2["sc"]), stream)
            assert np.array_equal(o_c2, g_s2), "G2 MSM sample mismatch vs oracle pippenger"
            thr2 = os.cpu_count() or 1
            rall2 = None
            mm2 = min(m2, n2 // thr2) if thr2 > 1 else 0
            if thr2 > 1 and mm2 >= 64:
                from concurrent.futures import ThreadPoolExecutor
                t0 = time.perf_counter()
                with ThreadPoolExecutor(max_workers=thr2) as ex:
                    list(ex.map(lambda t: cport.pippenger("bls12_381_g2", pts2_h[t * mm2:(t + 1) * mm2], sc2_h[t * mm2:(t + 1) * mm2]), range(thr2)))
                rall2 = thr2 * mm2 / (time.perf_counter() - t0)
            port_e2 = baseline_entry(m2 / dt2, m2, rall2, thr2 * mm2 if rall2 else 0, thr2, "points/s",
                                     "first %d points of the same MSM through oracle/c pippenger over Fp2 (curve.ts:863-905, "
                                     "tower.ts:393-475 restated), result compared bit-exactly with the GPU MSM on the same subset; "
                                     "all_threads: one independent MSM of n/threads points per thread (points/s summed)")
            ref_info2 = None
            if ref_ok:
                mr2 = min(n2, 1 << 10)
                o_r2, ref_info2 = refjs.pippenger(BLS12_381_G2, pts2_h[:mr2], sc2_h[:mr2])
                g_r2, _ = eng.msm_dev(BLS12_381_G2, mr2, dev_ptr(sub2["pts"]), dev_ptr(sub2["sc"]), stream)
                assert np.array_equal(o_r2, g_r2), "the reference's G2 pippenger differs from the GPU MSM"
            msm2["cpu_baseline"] = with_reference(port_e2, ref_info2, "points/s",
                                                  "first 1024 points of the same MSM through the reference's pippenger over Fp2, one call; result compared "
                                                  "bit-exactly with the GPU MSM on the same subset")
        extra["msm_g2"] = msm2
        if dist_on:
            ns = n2 // world
            sub_expect = sum(k * p for k, p in zip(sub2["ks"][:ns], sub2["pks"][:ns])) % BLS_R
            hs2 = {}

            def step_strong2():
                hs2["r"] = msm_sharded(eng, BLS12_381_G2, ns, dev_ptr(sub2["pts"]), dev_ptr(sub2["sc"]), stream, device, n_max=ns)

            st_s2 = time_steps(step_strong2, K, W, dist_on)
            wall_s, _ = st_s2
            wall_s = max_over_ranks(wall_s, dist_on, device)
            tot = sum_over_ranks_bigint(sub_expect, BLS_R, dist_on, device)
            got_s, _ = hs2["r"]
            assert wire_to_affine(BLS12_381_G2, got_s) == BlsG2.BASE.multiplyUnsafe(tot).toAffine(), "strong G2 MSM mismatch"
            by_points2 = {"metric": "bls12_381_g2_msm_points_per_sec", "value": ns * world * K / wall_s,
                          "unit": "points/s", "ms_per_msm": wall_s / K * 1e3, "total_points": ns * world,
                          "points_per_gpu": ns, "scaling": "strong", "mode": "points", "step_times": st_s2.dist()}
            extra["msm_g2_strong"] = strong_by_windows(BLS12_381_G2, BlsG2, n2, 0x6D736D0000001004)
            extra["msm_g2_strong"]["by_points"] = by_points2
        if not result:
            result = dict(msm2)
            result.update({"n_gpus": world, "steps": K, "warmup": W, "ms_per_step": msm2["ms_per_msm"],
                           "higher_is_better": True, "vs_baseline": None, "dtype": "u32",
                           "data": "synthetic: P_i=(a+i*b)G2, s_i uniform in [0,2^254), every 17th zero",
                           "config": {"workload": "bls12-381 G2 Pippenger MSM, 2^%d points per GPU" % (args.log2n - 2)}})
            extra.pop("msm_g2", None)

    # ------------------------------------------------------------------ ed25519 batch verify (configs[2])
    if args.workload in ("all", "ed25519"):
        import hashlib
        from helpers import load_golden
        from oracle.curves import ED25519_L, Ed25519
        from oracle.edwards import eddsa_verify
        nv = max(64, n >> 2)                                  # 2^18 at the default size
        eb = make_ed25519_batch(eng, nv, rank, device, stream)
        sig_np, pk_np, msg_list, expect, tail, nz = eb["sig"], eb["pk"], eb["msgs"], eb["expect"], eb["tail"], eb["nz"]
        d_sig, d_pk, d_blob, d_off = eb["d_sig"], eb["d_pk"], eb["d_blob"], eb["d_off"]
        d_ok = torch.empty((nv,), dtype=torch.uint8, device=device)
        d_k2 = torch.empty((nv, 32), dtype=torch.uint8, device=device)

        # kernel-only rate first (pre-hashed challenges, the r01 figure): it also brings the clocks to their steady
        # state before the hash-inclusive measurement that the headline of this workload quotes
        eng.ed25519_challenge_batch_dev(nv, dev_ptr(d_sig), dev_ptr(d_pk), dev_ptr(d_blob), dev_ptr(d_off), dev_ptr(d_k2), stream)

        def step_ed_k():
            eng.ed25519_verify_batch_dev(nv, dev_ptr(d_sig), dev_ptr(d_pk), dev_ptr(d_k2), True, dev_ptr(d_ok), stream)

        st_edk = time_steps(step_ed_k, K, W, dist_on)
        wall_k, ev_ms_k = st_edk
        wall_k = max_over_ranks(wall_k, dist_on, device)
        assert np.array_equal(d_ok.cpu().numpy().astype(bool), expect), "ed25519 verdict mismatch (pre-hashed)"

        def step_ed():     # hash on the device + verify: the reference's verify() from (sig, msg, pk)
            eng.ed25519_verify_batch_msgs_dev(nv, dev_ptr(d_sig), dev_ptr(d_pk), dev_ptr(d_blob), dev_ptr(d_off), True, dev_ptr(d_ok), stream)

        st_ed = time_steps(step_ed, K, W, dist_on)
        wall, ev_ms = st_ed
        pstate_e = power_state(step_ed, dev_index, 0.9) if (rank == 0 and world == 1 and not args.quick_verify) else None
        wall = max_over_ranks(wall, dist_on, device)
        got = d_ok.cpu().numpy().astype(bool)
        assert np.array_equal(got, expect), "ed25519 verdict mismatch vs construction / zip215.json"
        # strict (RFC 8032) mode once, untimed: construction for the synthetic part, the oracle for the 196 cases
        eng.ed25519_verify_batch_msgs_dev(nv, dev_ptr(d_sig), dev_ptr(d_pk), dev_ptr(d_blob), dev_ptr(d_off), False, dev_ptr(d_ok), stream)
        torch.cuda.synchronize()
        got_strict = d_ok.cpu().numpy().astype(bool)
        assert np.array_equal(got_strict[:tail], expect[:tail]), "ed25519 strict-mode verdict mismatch"
        if not args.quick_verify:
            for j in range(nz):
                assert got_strict[tail + j] == eddsa_verify(Ed25519, sig_np[tail + j].tobytes(), b"Zcash", pk_np[tail + j].tobytes(), zip215=False), \
                    "strict-mode verdict differs from the oracle on zip215.json case %d" % j
            for i in list(range(0, 24)) + list(range(63, tail, max(64, tail // 16 // 64 * 64))):
                assert got[i] == eddsa_verify(Ed25519, sig_np[i].tobytes(), msg_list[i], pk_np[i].tobytes(), zip215=True)
        ed_cpu = None
        if cpu_leg:
            k2_np = d_k2.cpu().numpy()

            def work(lo, hi):
                for i in range(lo, hi):                        # the reference's verify hashes too
                    hashlib.sha512(sig_np[i, :32].tobytes() + pk_np[i].tobytes() + msg_list[i]).digest()
                return cport.ed25519_verify_batch(sig_np[lo:hi], pk_np[lo:hi], k2_np[lo:hi], True)

            def check(lo, hi, r):
                assert np.array_equal(r, got[lo:hi]), "ed25519 sample mismatch vs oracle/c"
            r1, d1, rall, dall, thr = cpu_baseline_rates(work, nv, 250, args.cpu_seconds, check)
            ed_cpu = baseline_entry(r1, d1, rall, dall, thr, "verifies/s",
                                    "first %d signatures of the same batch through oracle/c (edwards.ts:942-989 restated) plus "
                                    "hashlib SHA-512 per item, verdicts compared with the GPU's")
            if ref_ok:
                mr = 2048
                pick = list(range(mr - 196)) + list(range(tail, tail + 196)) if nz == 196 and tail + 196 <= nv else list(range(mr))
                v_r, ed_ref = refjs.ed25519_verify([sig_np[i].tobytes() for i in pick], [msg_list[i] for i in pick],
                                                   [pk_np[i].tobytes() for i in pick], zip215=True)
                assert np.array_equal(v_r, got[pick]), "the reference's ed25519.verify differs from the GPU verdicts"
                ed_cpu = with_reference(ed_cpu, ed_ref, "verifies/s",
                                        "2048 signatures of the same batch (incl. the 196 zip215.json cases) through the reference's ed25519.verify "
                                        "(edwards.ts:942-989, SHA-512 by node:crypto), verdicts compared with the GPU's")
        kern_s = ev_ms_k / K * 1e-3
        traffic, tsrc = pmc.traffic("ed25519") if nv == 1 << 18 else (None, None)
        extra["ed25519_verify"] = {"metric": "ed25519_verifies_per_sec", "value": world * nv * K / wall,
                                   "unit": "verifies/s", "ms_per_batch": wall / K * 1e3, "sigs_per_gpu": nv,
                                   "step_times": st_ed.dist(),
                                   "kernel_only": {"value": world * nv * K / wall_k, "ms_per_batch": wall_k / K * 1e3,
                                                   "step_times": st_edk.dist(),
                                                   "note": "pre-hashed challenges (ncg_ed25519_verify_batch_dev)"},
                                   "note": "verify from (sig, msg, pk): SHA-512(R||A||M) mod L and the curve arithmetic both on the device "
                                           "(ncg_ed25519_verify_batch_msgs_dev); %d distinct key pairs, 32-byte messages, 1/64 corrupted "
                                           "(R, s or message), the reference's %d zip215.json cases appended; zip215 = true timed, "
                                           "strict mode verified once" % (tail, nz),
                                   "roofline": {"bound": "valu", "bound_note": BOUND_NOTE, "achieved": 161.0 * nv / (ev_ms / K * 1e-3) / 1e9,
                                                "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                "frac": 161.0 * nv / (ev_ms / K * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                "traffic": traffic, "traffic_source": tsrc,
                                                "kernel": "k_ed25519_verify (+ k_ed25519_challenge in the hash-inclusive figure)",
                                                "kernel_ms": ev_ms / K, "kernel_only_ms": ev_ms_k / K,
                                                "valu": valu_block(pmc, "ed25519", kern_s, 4.9e5 * nv, ed25519_mads_per_verify() * nv)}}
        if pstate_e:
            pstate_e["ms_per_batch_x_sclk"] = wall / K * 1e3 * pstate_e["sclk_mhz"][1]
            extra["ed25519_verify"]["power_state"] = pstate_e
        if ed_cpu:
            extra["ed25519_verify"]["cpu_baseline"] = ed_cpu

    # ------------------------------------------------------------------ NTT over Fr
    if args.workload in ("all", "ntt"):
        from noble_curves_amd import fft as gfft
        bits = min(22, args.log2n + 2)                         # 2^22 coefficients per GPU at the default size
        nn = 1 << bits
        roots = gfft.rootsOfUnity(gfft.bls12_381_Fr, 7)
        om = roots.omega(bits)
        gen = torch.Generator(device=device)
        gen.manual_seed(0x4E77 + rank)
        x = torch.randint(0, 256, (nn, 32), dtype=torch.uint8, device=device, generator=gen)
        x[:, 31] &= 0x3F                                       # < 2^254 < r: canonical residues
        y = torch.empty_like(x)
        z = torch.empty_like(x)

        def step_ntt():
            eng.ntt_dev(bits, 1, om, dev_ptr(x), dev_ptr(y), stream)

        st_ntt = time_steps(step_ntt, K, W, dist_on)
        wall, ev_ms = st_ntt
        wall = max_over_ranks(wall, dist_on, device)
        eng.ntt_dev(bits, 1, om, dev_ptr(y), dev_ptr(z), stream, inverse=True)
        torch.cuda.synchronize()
        assert bool((z == x).all().item()), "NTT: inverse(direct(x)) != x"
        xs = x.cpu().numpy()
        le = xs.view("<u8").reshape(nn, 4).astype(object)
        acc = (int(le[:, 0].sum()) + (int(le[:, 1].sum()) << 64) + (int(le[:, 2].sum()) << 128) + (int(le[:, 3].sum()) << 192)) % BLS_R
        assert int.from_bytes(y[0].cpu().numpy().tobytes(), "little") == acc, "NTT: y[0] != sum of inputs"
        sb = 12
        ys = eng.ntt(sb, xs[:1 << sb], roots.omega(sb))
        assert np.array_equal(ys, cport.fft_fr(sb, xs[:1 << sb], roots.omega(sb))), "NTT sample mismatch vs oracle/c"
        ntt_cpu = None
        if cpu_leg:
            cb = 18
            t0 = time.perf_counter()
            reps = 0
            while time.perf_counter() - t0 < min(args.cpu_seconds, 6.0):
                yc = cport.fft_fr(cb, xs[:1 << cb], roots.omega(cb))
                reps += 1
            dt = time.perf_counter() - t0
            assert np.array_equal(yc, eng.ntt(cb, xs[:1 << cb], roots.omega(cb))), "NTT 2^18 mismatch vs oracle/c"
            ntt_cpu = baseline_entry(reps * (1 << cb) / dt, reps, None, 0, 1, "elements/s",
                                     "%%d transforms of 2^%d of the same coefficients through oracle/c (fft.ts:422-480 loop "
                                     "restated; includes its roots-table build), output compared with the GPU's" % cb)
        npass = 3 if bits == 22 else None
        traffic, tsrc = pmc.traffic("ntt", npass) if npass else (None, None)
        extra["ntt_fr"] = {"metric": "bls12_381_fr_ntt_elements_per_sec", "value": world * nn * K / wall,
                           "unit": "elements/s", "ms_per_transform": wall / K * 1e3, "log2n": bits,
                           "step_times": st_ntt.dist(),
                           "note": "FFT(roots, Fr).direct, natural in / natural out, one 2^%d transform per GPU" % bits,
                           "roofline": {"bound": "valu", "bound_note": BOUND_NOTE, "achieved": 64.0 * nn / (ev_ms / K * 1e-3) / 1e9,
                                        "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                        "frac": 64.0 * nn / (ev_ms / K * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                        "traffic": traffic, "traffic_source": tsrc,
                                        "kernel": "k_ntt_pass (3 launches per 2^22 transform; achieved and traffic are per transform: "
                                                  "3 x the average launch)",
                                        "kernel_ms": ev_ms / K,
                                        # fr29.hpp: 81 + 72 + 9 multiply-adds per twiddle product (the first stage has none), 18 per
                                        # element and pass for the fold below 2^256; the reference does one Fr.mul per butterfly
                                        "valu": valu_block(pmc, "ntt", ev_ms / K * 1e-3, 136.0 * (nn / 2 * bits),
                                                           162.0 * (nn / 2 * (bits - 1)) + 18.0 * nn * (npass or 1),
                                                           launches=npass or 1)}}
        if ntt_cpu:
            extra["ntt_fr"]["cpu_baseline"] = ntt_cpu

    if not result and extra:   # single-workload runs of the rows that have no headline line of their own
        key = next(iter(extra))
        e = extra.pop(key)
        result = dict(e)
        result.update({"n_gpus": world, "steps": K, "warmup": W, "higher_is_better": True, "vs_baseline": None, "dtype": "u32",
                       "ms_per_step": e.get("ms_per_batch", e.get("ms_per_transform")), "data": "synthetic",
                       "scaling": "weak", "config": {"workload": key}})
    if extra:
        result["extra"] = extra
    if rank == 0:
        result["host"] = host
        result["prewarm"] = (("%d untimed steps" % PREWARM_DIST_STEPS if dist_on else "%.0f ms of untimed steps" % (PREWARM_S * 1e3)) +
                             " before the W warm-up steps of every timed loop (boost-clock settling; the K timed steps are unchanged)")
        result["pmc"] = "live rocprofv3 passes in this run" if live else "committed profile (see roofline.traffic_source)"
        if args.dist_dry_run:
            result["dist_dry_run"] = "multi-rank code path executed with a world of one rank (nccl group + the engine's RCCL communicator); not a scaling figure"
        # stdout carries ONE compact line (tools/bench_compact.py, < 6 KB: the driver parses the last line of an 8 KB tail -
        # VERDICT r04 #1); the full object - step lists, notes, every sub-measurement - goes to --out
        full_path = args.out or os.path.join(ROOT, "gpurun_out", "bench_full.json")
        try:
            os.makedirs(os.path.dirname(os.path.abspath(full_path)), exist_ok=True)
            with open(full_path, "w") as f:
                f.write(json.dumps(result) + "\n")
        except OSError:
            full_path = None
        import bench_compact
        json_out.write(bench_compact.dumps(bench_compact.compact_line(result, full_path and os.path.relpath(full_path, ROOT))) + "\n")
        json_out.flush()
    if dist_on:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
