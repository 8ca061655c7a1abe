// HOST-ONLY unit-test shim: runs the same __host__ __device__ templates the kernels use
// (field arithmetic, group law, GLV split, window recoding, per-lane ladders) on the CPU so
// their logic can be checked against the oracle without a GPU (`pytest -m "not gpu"`).
// Built into tests/_build/libncg_hosttest.so; never linked into libncg.so, never a fallback.
#include <vector>

#include "mulvar.hpp"
#include "ed25519.hip"  // single-TU inclusion: lane function + host table builder
#include "decode.hip"
#include "ntt.hip"
#include "h2c.hip"
#include "endo.hpp"
#include "ecdsa.hip"
#include "msm_shard.hpp"

using namespace ncg;

template <class C, int W>
static void ht_mul_var_t(const uint32_t* pts, const uint32_t* scalars, uint32_t* out, uint8_t* out_inf, int n) {
  constexpr int FW = MulVarCfg<C, W>::FW, WW = MulVarCfg<C, W>::WW;
  std::vector<uint32_t> tab(MulVarCfg<C, W>::TS * 2 * FW);
  for (int i = 0; i < n; i++)
    mul_var_lane<C, W>(pts + (size_t)i * 2 * WW, scalars + (size_t)i * 8, out + (size_t)i * 2 * WW, out_inf + i, true,
                       tab.data(), 1);
}

template <class PR>
static void ht_field_t(int op, const uint32_t* a, const uint32_t* b, uint32_t* r) {
  Fp<PR> x = fp_to_mont<PR>(fp_load<PR>(a)), y = fp_to_mont<PR>(fp_load<PR>(b)), z;
  switch (op) {
    case 0: z = x * y; break;
    case 1: z = fp_sqr<PR>(x); break;
    case 2: z = x + y; break;
    case 3: z = x - y; break;
    case 4: z = fp_neg<PR>(x); break;
    case 5: z = fp_inv<PR>(x); break;
    default: z = Fp<PR>::zero();
  }
  fp_store<PR>(r, fp_from_mont<PR>(z));
}

// Fe9 ops on RAW limb arrays (so tests can feed the loosest limbs each bound type admits).
// variant picks the operand bounds (A, B); the result is written as the canonical wire value.
template <class PR, int A, int B>
static void ht_fe9_ab(int op, const uint32_t* a, const uint32_t* b, uint32_t* r) {
  Fe9<PR, A> x;
  Fe9<PR, B> y;
  for (int i = 0; i < 9; i++) {
    x.v[i] = a[i];
    y.v[i] = b[i];
  }
  switch (op) {
    case 0: fe9_to_wire(r, x * y); break;
    case 1: if constexpr (A <= 2) fe9_to_wire(r, f_sqr(x)); else fe9_to_wire(r, f_sqr(fe9_norm(x))); break;
    case 2: if constexpr (A + B <= 7) fe9_to_wire(r, x + y); break;
    case 3: if constexpr (A + B + 1 <= 7) fe9_to_wire(r, x - y); break;
    case 4: if constexpr (A + 1 <= 7) fe9_to_wire(r, f_neg(x)); break;
    case 5: fe9_to_wire(r, f_inv(x)); break;
    case 6: fe9_to_wire(r, fe9_norm(x)); break;
    case 7: r[0] = f_eqz(x) ? 1u : 0u; break;
    case 8: {  // norm must leave limbs below U
      auto n = fe9_norm(x);
      uint32_t mx = 0;
      for (int i = 0; i < 9; i++) mx = n.v[i] > mx ? n.v[i] : mx;
      r[0] = mx;
      break;
    }
    case 9: {  // product limbs (raw), to check the output bound
      auto n = x * y;
      for (int i = 0; i < 8; i++) r[i] = 0;
      uint32_t mx = 0;
      for (int i = 0; i < 9; i++) mx = n.v[i] > mx ? n.v[i] : mx;
      r[0] = mx;
      break;
    }
    case 10: if constexpr (2 * A <= 7) fe9_to_wire(r, f_dbl(x)); break;
  }
}
template <class PR>
static int ht_fe9_t(int op, int variant, const uint32_t* a, const uint32_t* b, uint32_t* r) {
  switch (variant) {
    case 11: ht_fe9_ab<PR, 1, 1>(op, a, b, r); return 0;
    case 12: ht_fe9_ab<PR, 1, 2>(op, a, b, r); return 0;
    case 17: ht_fe9_ab<PR, 1, 7>(op, a, b, r); return 0;
    case 71: ht_fe9_ab<PR, 7, 1>(op, a, b, r); return 0;
    case 23: ht_fe9_ab<PR, 2, 3>(op, a, b, r); return 0;
    case 32: ht_fe9_ab<PR, 3, 2>(op, a, b, r); return 0;
    case 22: ht_fe9_ab<PR, 2, 2>(op, a, b, r); return 0;
    case 15: ht_fe9_ab<PR, 1, 5>(op, a, b, r); return 0;
    case 33: ht_fe9_ab<PR, 3, 3>(op, a, b, r); return 0;
    case 77: ht_fe9_ab<PR, 7, 7>(op, a, b, r); return 0;
    case 46: ht_fe9_ab<PR, 4, 6>(op, a, b, r); return 0;
  }
  return -1;
}


// ---- sharded MSM twin (tests/test_distributed_cpu.py): the per-shard grouped window sums computed naively with the
// group-law templates, then the REAL slot format, header check, partial-sum order and finish of comm.hip / msm_finish.hpp
template <class C>
static int ht_shard_local_t(int curve, int n_local, int n_max, const uint32_t* pts_wire, const uint32_t* scalars, uint8_t* slot, int mode,
                            int part, int nparts) {
  using G = MsmGroup<C>;
  constexpr int XW = G::ACC_WORDS;
  MsmPlan pl;
  if (msm_make_plan_impl(curve, n_max, 0, &pl) != 0) return -1;
  const int ng = msm_ngroups(pl.c);
  int w0 = 0, cnt = pl.nwin;
  if (mode != SHARD_POINTS) msm_shard_window_range(pl.nwin, part, nparts, &w0, &cnt);
  // SHARD_WINDOWS_SHARED (a precomputed set: every window adds into ONE bucket set, msm.hpp): the slot is one grouped-sum array
  const size_t fin_words = (size_t)ng * (mode == SHARD_WINDOWS_SHARED ? 1 : cnt) * XW;
  memset(slot, 0, msm_shard_slot_bytes(curve));
  FinHeader h{(uint32_t)pl.c, (uint32_t)pl.nwin, (uint32_t)fin_words, (uint32_t)curve, (uint32_t)w0, (uint32_t)cnt, (uint32_t)mode, SHARD_NO_BAD};
  uint32_t* fin = (uint32_t*)(slot + sizeof h);
  std::vector<typename G::Acc> win(pl.nwin, G::identity());
  std::vector<uint32_t> st(G::AFF_WORDS);
  const uint32_t mask = (1u << pl.c) - 1u;
  const int half = 1 << (pl.c - 1);
  for (int i = 0; i < n_local; i++) {
    {  // scalar >= group order: the smallest such index travels in the header (k_msm_digits)
      uint32_t bw = 0;
      for (int j = 0; j < 8; j++) (void)__builtin_subc(scalars[(size_t)i * 8 + j], pl.order[j], bw, &bw);
      if (bw == 0 && h.bad == SHARD_NO_BAD) h.bad = (uint32_t)i;
    }
    G::wire_to_storage(pts_wire + (size_t)i * G::WIRE_AFF, st.data());
    const typename G::Aff P = G::aff_load(st.data());
    uint32_t my[11];
    uint32_t cy = 0;
    for (int j = 0; j < 8; j++) my[j] = __builtin_addc(scalars[(size_t)i * 8 + j], pl.hconst[j], cy, &cy);
    my[8] = __builtin_addc(0u, pl.hconst[8], cy, &cy);
    my[9] = pl.hconst[9] + cy;
    my[10] = 0;
    for (int w = w0; w < w0 + cnt; w++) {
      const int bp = w * pl.c, limb = bp >> 5, sft = bp & 31;
      const uint64_t two = ((uint64_t)my[limb + 1] << 32) | my[limb];
      const int d = (int)((uint32_t)(two >> sft) & mask) - half;
      if (d == 0) continue;
      const unsigned a = (unsigned)(d < 0 ? -d : d);
      typename G::Acc t = G::identity();
      for (int bit = 15; bit >= 0; bit--) {
        t = G::dbl(t);
        if ((a >> bit) & 1u) t = G::madd(t, P, d < 0);
      }
      win[w] = G::add(win[w], t);
    }
  }
  memcpy(slot, &h, sizeof h);
  if (mode == SHARD_WINDOWS_SHARED) {  // what the shifted copies 2^(c w) P make of this rank's windows: V_0 = sum_w 2^(c w) W_w
    typename G::Acc tot = G::identity();
    for (int w = w0 + cnt - 1; w >= w0; w--) {
      typename G::Acc t = win[w];
      for (int b = 0; b < pl.c * w; b++) t = G::dbl(t);
      tot = G::add(tot, t);
    }
    G::acc_store(fin, tot);  // V_j, j >= 1: all-zero words, as below
    return 0;
  }
  for (int w = 0; w < cnt; w++) G::acc_store(fin + (size_t)w * XW, win[w0 + w]);  // V_0 = W_w; V_j = identity for j >= 1
  return 0;
}
// the host half of comm.hip's job_finish_host around a naive term-by-term sum: header check, the scalar verdict of all
// ranks, window assembly (SHARD_WINDOWS) or the sum of the slots (SHARD_POINTS), host Horner
template <class C>
static int ht_shard_combine_t(int curve, int n_max, int nparts, const uint8_t* slots, uint32_t* out, uint8_t* out_inf, char* err, int errlen) {
  using G = MsmGroup<C>;
  constexpr int XW = G::ACC_WORDS;
  MsmPlan pl;
  if (msm_make_plan_impl(curve, n_max, 0, &pl) != 0) return -1;
  const size_t stride = msm_shard_slot_bytes(curve);
  std::vector<FinHeader> hs(nparts);
  for (int r = 0; r < nparts; r++) memcpy(&hs[r], slots + stride * r, sizeof(FinHeader));
  const uint32_t mode = hs[0].mode;
  char msg[400];
  if (msm_shard_check(hs.data(), nparts, curve, pl, mode, msg, sizeof msg) >= 0) {
    if (err && errlen > 0) snprintf(err, errlen, "%s", msg);
    return 1;
  }
  uint32_t bad_idx = 0;
  const int bad_rank = msm_shard_first_bad(hs.data(), nparts, &bad_idx);
  if (bad_rank >= 0) {
    if (err && errlen > 0) snprintf(err, errlen, "noble-gpu: msm_sharded: invalid scalar at index %u of shard %d (not below the group order)", bad_idx, bad_rank);
    return 2;
  }
  const int fin_nwin = mode == SHARD_WINDOWS_SHARED ? 1 : pl.nwin;   // comm.hip job_finish_host: one window, no Horner across windows
  const size_t fin_words = (size_t)msm_ngroups(pl.c) * fin_nwin * XW;
  std::vector<uint32_t> sum(fin_words);
  if (mode == SHARD_WINDOWS) {
    msm_shard_assemble_windows(slots, stride, hs.data(), nparts, curve, pl, sum.data());
  } else {
    const size_t npoints = fin_words / XW;
    for (size_t t = 0; t < npoints; t++) {
      typename G::Acc acc = G::acc_load((const uint32_t*)(slots + sizeof(FinHeader)) + t * XW);
      for (int r = 1; r < nparts; r++) acc = G::add(acc, G::acc_load((const uint32_t*)(slots + stride * r + sizeof(FinHeader)) + t * XW));
      G::acc_store(sum.data() + t * XW, acc);
    }
  }
  msm_host_finish_any<C>(sum.data(), pl.c, fin_nwin, out, out_inf);
  return 0;
}
#define HT_CURVE_DISPATCH(curve, CALL)                \
  switch (curve) {                                    \
    case CURVE_SECP256K1: return CALL(CurveSecp);     \
    case CURVE_ED25519: return CALL(CurveEd);         \
    case CURVE_BLS12_381_G1: return CALL(CurveG1);    \
    case CURVE_BLS12_381_G2: return CALL(CurveG2);    \
    default: return -1;                               \
  }

extern "C" {

// field: 0 secp256k1 p, 1 ed25519 p (radix-2^29 lazy form, fe9.hpp); a, b: 9 raw limbs; r: 8 wire words
int ht_fe9_op(int field, int op, int variant, const uint32_t* a, const uint32_t* b, uint32_t* r) {
  if (field == 0) return ht_fe9_t<Fe9SecpPR>(op, variant, a, b, r);
  if (field == 1) return ht_fe9_t<Fe9EdPR>(op, variant, a, b, r);
  return -1;
}

int ht_mul_var(int curve, const uint32_t* pts, const uint32_t* scalars, uint32_t* out, uint8_t* out_inf, int n) {
  switch (curve) {
    case CURVE_SECP256K1: ht_mul_var_t<CurveSecp, 4>(pts, scalars, out, out_inf, n); return 0;
    case CURVE_BLS12_381_G1: ht_mul_var_t<CurveG1, 3>(pts, scalars, out, out_inf, n); return 0;
    case 12: ht_mul_var_t<CurveG1E, 4>(pts, scalars, out, out_inf, n); return 0;  // subgroup points only (GLV ladder)
    case CURVE_BLS12_381_G2: ht_mul_var_t<CurveG2, 3>(pts, scalars, out, out_inf, n); return 0;
  }
  return -1;
}

// field: 0 secp256k1 p, 1 ed25519 p, 2 bls12-381 p; canonical LE limbs in and out
int ht_field_op(int field, int op, const uint32_t* a, const uint32_t* b, uint32_t* r) {
  switch (field) {
    case 0: ht_field_t<ParamsSecpP>(op, a, b, r); return 0;
    case 1: ht_field_t<ParamsEdP>(op, a, b, r); return 0;
    case 2: ht_field_t<ParamsBlsP>(op, a, b, r); return 0;
  }
  return -1;
}

// GLV split of a 256-bit scalar: out = k1[5] k2[5] k1neg k2neg (12 words)
int ht_glv_split(const uint32_t* k, uint32_t* out) {
  uint32_t kk[8];
  for (int i = 0; i < 8; i++) kk[i] = k[i];
  GlvSplit s = secp_glv_split(kk);
  for (int i = 0; i < 5; i++) {
    out[i] = s.k1[i];
    out[5 + i] = s.k2[i];
  }
  out[10] = s.k1neg;
  out[11] = s.k2neg;
  return 0;
}

int ht_ed25519_mul_var(const uint32_t* pts, const uint32_t* scalars, uint32_t* out, uint8_t* out_inf, int n) {
  for (int i = 0; i < n; i++) ed25519_mul_var_host(pts + (size_t)i * 16, scalars + (size_t)i * 8, out + (size_t)i * 16, out_inf + i);
  return 0;
}

// Fp2 square root lane: in = c0 c1 wire (24 words); out = root wire; returns 1 if a root exists
int ht_fp2_sqrt(const uint32_t* in, uint32_t* out) {
  Fe29x2<2> a{fe29_from_wire(in), fe29_from_wire(in + 12)}, r;
  bool ok = fe29x2_sqrt(a, r);
  fe29_to_wire(out, r.c0);
  fe29_to_wire(out + 12, r.c1);
  return ok ? 1 : 0;
}

int ht_decode_points(int curve, const uint8_t* in, int flags, uint32_t* out, uint8_t* ok, uint8_t* inf, int n) {
  decode_points_host(curve, in, flags, out, ok, inf, n);
  return 0;
}

int ht_encode_points(int curve, const uint32_t* in, uint8_t* out, uint8_t* ok, int n) {
  encode_points_host(curve, in, out, ok, n);
  return 0;
}

int ht_map_to_curve(int curve, const uint32_t* u, int count, uint32_t* out, uint8_t* inf, int n) {
  map_to_curve_host(curve, u, count, out, inf, n);
  return 0;
}

// NTT over Fr on the CPU through the device field code (butterflies + table walk of ntt.hip)
int ht_ntt(int n, const uint32_t* omega, const uint32_t* in, uint32_t* out, int flags) {
  return ntt_host(n, omega, in, out, flags);  // number of column / limb overflows seen by the host checks
}
// the same with smaller passes (first pass of at most t0max stages, the others of at most tmax)
int ht_ntt_small_passes(int n, const uint32_t* omega, const uint32_t* in, uint32_t* out, int flags, int t0max, int tmax) {
  if (t0max < 1 || tmax < 1 || 1 + (n - (n < t0max ? n : t0max) + tmax - 1) / tmax > 8) return -1;
  return ntt_host(n, omega, in, out, flags, t0max, tmax);
}
// fr29.hpp on RAW limbs (9 words each).  op 0: mont(a, b) -> 9 limbs; 1: a + b; 2: a + 3r - b; 3: weak(a);
// 4: reduce256(a); 5: cond_sub(a); 6: from_words(a[0..7]); 7: to_words(a) -> 8 words.
// Returns the overflow count of the call.
int ht_fr29_op(int op, const uint32_t* a, const uint32_t* b, uint32_t* r) {
  Fr29 x, y, z;
  for (int i = 0; i < 9; i++) {
    x.v[i] = a[i];
    y.v[i] = b ? b[i] : 0;
    z.v[i] = 0;
  }
  fr29_overflows() = 0;
  switch (op) {
    case 0: z = fr29_mont(x, y); break;
    case 1: z = fr29_add(x, y); break;
    case 2: z = fr29_sub(x, y); break;
    case 3: z = fr29_weak(x); break;
    case 4: z = fr29_reduce256(x); break;
    case 5: z = fr29_cond_sub(x); break;
    case 6: {
      uint32_t w[8];
      for (int i = 0; i < 8; i++) w[i] = a[i];
      z = fr29_from_words(w);
      break;
    }
    case 7: {
      uint32_t w[8];
      fr29_to_words(w, x);
      for (int i = 0; i < 8; i++) z.v[i] = w[i];
      z.v[8] = 0;
      break;
    }
    default: return -1;
  }
  for (int i = 0; i < 9; i++) r[i] = z.v[i];
  return fr29_overflows();
}
// the pass planner of ntt_run: writes s_lo/T pairs, returns the number of passes
int ht_ntt_plan(int n, int* out) {
  int s_lo[8], T[8];
  int np = ntt_plan(n, s_lo, T);
  for (int i = 0; i < np; i++) {
    out[2 * i] = s_lo[i];
    out[2 * i + 1] = T[i];
  }
  return np;
}

// challenge scalar k = SHA-512(R || A || M) mod L through the device code (sha512.hpp); out: 8 words
int ht_ed25519_challenge(const uint8_t* sig, const uint8_t* pk, const uint8_t* msg, uint64_t len, uint32_t* out) {
  ed25519_challenge_host(sig, pk, msg, len, out);
  return 0;
}

// bls12-381 scalar split of the endomorphism MSM (endo.hpp): E = 2 (G1) or 4 (G2) sub-scalars, 6 words each
// (192-bit two's complement)
int ht_bls_endo_split(int E, const uint32_t* k8, uint32_t* out) {
  uint32_t k[8];
  for (int i = 0; i < 8; i++) k[i] = k8[i];
  if (E == 2) {
    uint32_t o[2][6];
    bls_endo_split2(o, k);
    for (int e = 0; e < 2; e++)
      for (int i = 0; i < 6; i++) out[e * 6 + i] = o[e][i];
    return 0;
  }
  if (E == 4) {
    uint32_t o[4][6];
    bls_endo_split4(o, k);
    for (int e = 0; e < 4; e++)
      for (int i = 0; i < 6; i++) out[e * 6 + i] = o[e][i];
    return 0;
  }
  return -1;
}

// ECDSA scalar side of one signature through the lane code (ecdsa.hip): u1 = h / s, u2 = r / s mod n, 8 LE words each
int ht_ecdsa_prepare(const uint8_t* sig64, const uint8_t* hash32, int low_s, uint32_t* u1, uint32_t* u2) {
  uint8_t ok = 0;
  ecdsa_prepare_host(sig64, hash32, low_s != 0, u1, u2, &ok);
  return ok;
}

// ed25519 verify of one item on the CPU through the kernel's lane function
int ht_ed25519_verify(const uint32_t* sig, const uint32_t* pk, const uint32_t* k, int zip215) {
  static uint32_t btab[ED25519_BTAB_WORDS];
  static bool built = false;
  if (!built) {
    ed25519_build_base_table(btab);
    built = true;
  }
  return ed25519_verify_host(sig, pk, k, btab, zip215 != 0) ? 1 : 0;
}

// the scalar side of the halved ed25519 verification (ed_halve.hpp): out = u[4] | v[4] | uneg | w[8], w = u s mod L
int ht_ed_halve(const uint32_t* k, const uint32_t* s, uint32_t* out) {
  uint32_t kk[8], ss[8], w[8];
  for (int i = 0; i < 8; i++) {
    kk[i] = k[i];
    ss[i] = s[i];
  }
  const EdHalf h = ed_halve_scalar(kk);
  ed_mul_mod_l(w, h.u, ss);
  for (int i = 0; i < 4; i++) {
    out[i] = h.u[i];
    out[4 + i] = h.v[i];
  }
  out[8] = h.uneg ? 1u : 0u;
  for (int i = 0; i < 8; i++) out[9 + i] = w[i];
  return 0;
}

size_t ht_msm_shard_slot_bytes(int curve) { return msm_shard_slot_bytes(curve); }
int ht_msm_shard_windows_local(int curve, int n, int part, int nparts, const uint32_t* pts_wire, const uint32_t* scalars, uint8_t* slot, int shared) {
#define CALL(C) ht_shard_local_t<C>(curve, n, n, pts_wire, scalars, slot, shared ? SHARD_WINDOWS_SHARED : SHARD_WINDOWS, part, nparts)
  HT_CURVE_DISPATCH(curve, CALL)
#undef CALL
}
int ht_msm_shard_local(int curve, int n_local, int n_max, const uint32_t* pts_wire, const uint32_t* scalars, uint8_t* slot) {
#define CALL(C) ht_shard_local_t<C>(curve, n_local, n_max, pts_wire, scalars, slot, SHARD_POINTS, 0, 1)
  HT_CURVE_DISPATCH(curve, CALL)
#undef CALL
}
// the lane segment msm_seg picks for a whole generic plan of n points (msm_plan.hpp): out = {c, nwin, nb, seg, nseg, lanes per window << ls, accum_waves}
int ht_msm_seg(int curve, int n, int c_override, int* out) {
  MsmPlan pl;
  if (msm_make_plan_impl(curve, n, c_override, &pl) != 0) return -1;
  const MsmSeg sg = msm_seg(pl);
  out[0] = pl.c; out[1] = pl.nwin; out[2] = pl.nb; out[3] = sg.seg; out[4] = sg.nseg; out[5] = sg.nseg << pl.ls; out[6] = pl.accum_waves;
  return 0;
}
int ht_msm_shard_combine(int curve, int n_max, int nparts, const uint8_t* slots, uint32_t* out, uint8_t* out_inf, char* err, int errlen) {
#define CALL(C) ht_shard_combine_t<C>(curve, n_max, nparts, slots, out, out_inf, err, errlen)
  HT_CURVE_DISPATCH(curve, CALL)
#undef CALL
}
// host finish of the MSM alone: variant 0 = the device templates compiled for the host, 1 = the default
// (bls12-381: 64-bit-limb Jacobian form of bls_host64.hpp; its field product on MULX / ADX where the CPU has them),
// 2 = the default with the portable field product forced, 3 = the default with the helper threads of the finish woken inside the
// call (bls_host64.hpp FinishPool: windows built by the helpers, the chain by the caller), 4 = the helper threads off
static int ht_msm_finish_default(int curve, int c, int nwin, const uint32_t* fin, uint32_t* out, uint8_t* out_inf) {
#define CALL(C) (msm_host_finish_any<C>(fin, c, nwin, out, out_inf), 0)
  HT_CURVE_DISPATCH(curve, CALL)
#undef CALL
}
int ht_msm_finish(int curve, int c, int nwin, const uint32_t* fin, uint32_t* out, uint8_t* out_inf, int variant) {
  if (variant == 0) {
#define CALL(C) (msm_host_finish<C>(fin, c, nwin, out, out_inf), 0)
    HT_CURVE_DISPATCH(curve, CALL)
#undef CALL
  }
  h64::adx_override() = variant == 2 ? 0 : -1;
  h64::finish_threads_override() = variant == 3 ? 2 : variant == 4 ? 0 : 1;
  const int rc = ht_msm_finish_default(curve, c, nwin, fin, out, out_inf);
  h64::adx_override() = -1;
  h64::finish_threads_override() = 1;
  return rc;
}
int ht_h64_have_adx(void) { return h64::have_adx() ? 1 : 0; }
// bls_host64.hpp on raw operands: op 0 / 2: Montgomery product of two canonical residues given as 6 x 64-bit words (result
// 6 words; 0 = as dispatched, 2 = the portable form); op 1: from_fe29 of 14 stored limbs (result 6 words, Montgomery form R = 2^384)
int ht_h64_op(int op, const uint64_t* a, const uint64_t* b, const uint32_t* limbs, uint64_t* r) {
  h64::Fp x, y, z;
  if (op == 0) {
    for (int i = 0; i < 6; i++) {
      x.v[i] = a[i];
      y.v[i] = b[i];
    }
    z = h64::mul(x, y);
  } else if (op == 2) {  // the portable product, whatever the CPU
    for (int i = 0; i < 6; i++) {
      x.v[i] = a[i];
      y.v[i] = b[i];
    }
    z = h64::mul_c(x, y);
  } else if (op == 1) {
    z = h64::from_fe29(limbs);
  } else {
    return -1;
  }
  for (int i = 0; i < 6; i++) r[i] = z.v[i];
  return 0;
}
int ht_msm_plan(int curve, int n, int* out) {  // c, nwin, ngroups, acc words
  MsmPlan pl;
  if (msm_make_plan_impl(curve, n, 0, &pl) != 0) return -1;
  out[0] = pl.c;
  out[1] = pl.nwin;
  out[2] = msm_ngroups(pl.c);
  out[3] = (int)msm_acc_words_inl(curve);
  return 0;
}

// the short-top-window part of a plan (MsmPlan::top_tb): c (forced when c_override > 0), nwin, top_tb, top_submask, the
// largest field the top window can hold (msm_plan_top_vmax) and H' (10 words) - tests/test_host_logic.py replays the digit
// kernel's mapping on them with Python integers
int ht_msm_plan_top(int curve, int n, int c_override, uint32_t* out16) {
  MsmPlan pl;
  if (msm_make_plan_impl(curve, n, c_override, &pl) != 0) return -1;
  out16[0] = (uint32_t)pl.c;
  out16[1] = (uint32_t)pl.nwin;
  out16[2] = (uint32_t)pl.top_tb;
  out16[3] = pl.top_submask;
  out16[4] = msm_plan_top_vmax(pl);
  out16[5] = (uint32_t)pl.nb;
  for (int i = 0; i < 10; i++) out16[6 + i] = pl.hconst[i];
  return 0;
}

}  // extern "C"
