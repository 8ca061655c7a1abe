/* C restatement of the reference's hot path for CPU timing and large-size cross-checks.
 * TEST INFRASTRUCTURE ONLY (see ../__init__.py): used by tests/ and by bench.py's
 * `cpu_baseline` leg ("kind": "port"); never linked into or called from the product path.
 *
 * Follows, function by function:
 *   secp256k1 Point.multiplyUnsafe  src/abstract/weierstrass.ts:915-928
 *     -> pushWnafPair :660-671, _splitEndoScalar :121-148 (divNearest :106),
 *        constants src/secp256k1.ts:48-64
 *     -> mulAddUnsafe src/abstract/curve.ts:820-836 (oddMultiples :420-425, wnafDigits :431-447,
 *        wnafWalk :479-498)
 *   bls12-381 G1 pippenger          src/abstract/curve.ts:863-905
 *   group law: RCB complete formulas  src/abstract/weierstrass.ts:793-880 (field_tmpl.h)
 * Parity: pinned against the Python oracle (itself pinned by the reference's fixtures) in
 * tests/test_oracle_c.py.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define NL 4
#define PFX(x) k1_##x
#include "field_tmpl.h"
#undef NL
#undef PFX
#define NL 6
#define PFX(x) bls_##x
#include "field_tmpl.h"
#undef NL
#undef PFX

/* ---- bls12-381 G2: the same group-law template over Fp2 = Fp[u] / (u^2 + 1) (src/abstract/tower.ts:393-475;
 * curve constants src/bls12-381.ts:321-345: b = 4 + 4u).  Wire: c0 || c1, 48 bytes each. */
static bls_ctx BLS;
typedef struct { bls_fe c0, c1; } g2_fe;
typedef struct { g2_fe r1, b3; } g2_ctx;
static g2_ctx G2C;
static inline int g2_is0(const g2_fe* a) { return bls_is0(&a->c0) && bls_is0(&a->c1); }
static inline void g2_add(const g2_ctx* c, g2_fe* r, const g2_fe* a, const g2_fe* b) { /* tower.ts:404 */
  (void)c;
  bls_add(&BLS, &r->c0, &a->c0, &b->c0);
  bls_add(&BLS, &r->c1, &a->c1, &b->c1);
}
static inline void g2_sub(const g2_ctx* c, g2_fe* r, const g2_fe* a, const g2_fe* b) { /* :413 */
  (void)c;
  bls_sub(&BLS, &r->c0, &a->c0, &b->c0);
  bls_sub(&BLS, &r->c1, &a->c1, &b->c1);
}
static inline void g2_neg(const g2_ctx* c, g2_fe* r, const g2_fe* a) { /* :393 */
  (void)c;
  bls_neg(&BLS, &r->c0, &a->c0);
  bls_neg(&BLS, &r->c1, &a->c1);
}
static inline void g2_mul(const g2_ctx* c, g2_fe* r, const g2_fe* a, const g2_fe* b) { /* :420-431, Karatsuba */
  (void)c;
  bls_fe t1, t2, s1, s2, m;
  bls_mul(&BLS, &t1, &a->c0, &b->c0);
  bls_mul(&BLS, &t2, &a->c1, &b->c1);
  bls_add(&BLS, &s1, &a->c0, &a->c1);
  bls_add(&BLS, &s2, &b->c0, &b->c1);
  bls_mul(&BLS, &m, &s1, &s2);
  bls_sub(&BLS, &r->c0, &t1, &t2);
  bls_add(&BLS, &t1, &t1, &t2);
  bls_sub(&BLS, &r->c1, &m, &t1);
}
static void g2_inv(const g2_ctx* c, g2_fe* r, const g2_fe* a) { /* :458-475 */
  (void)c;
  bls_fe n0, n1, f;
  bls_mul(&BLS, &n0, &a->c0, &a->c0);
  bls_mul(&BLS, &n1, &a->c1, &a->c1);
  bls_add(&BLS, &n0, &n0, &n1);
  bls_inv(&BLS, &f, &n0);
  bls_mul(&BLS, &r->c0, &f, &a->c0);
  bls_neg(&BLS, &n1, &a->c1);
  bls_mul(&BLS, &r->c1, &f, &n1);
}
static void g2_tomont(const g2_ctx* c, g2_fe* r, const g2_fe* a) {
  (void)c;
  bls_tomont(&BLS, &r->c0, &a->c0);
  bls_tomont(&BLS, &r->c1, &a->c1);
}
static void g2_frommont(const g2_ctx* c, g2_fe* r, const g2_fe* a) {
  (void)c;
  bls_frommont(&BLS, &r->c0, &a->c0);
  bls_frommont(&BLS, &r->c1, &a->c1);
}
#define NL 12
#define PFX(x) g2_##x
#define PT_ONLY
#include "field_tmpl.h"
#undef PT_ONLY
#undef NL
#undef PFX

/* ---- small multi-limb helpers (32-bit limbs, little endian) for the GLV split ---- */
#define MPN 16
typedef struct { uint32_t w[MPN]; int neg; } mp; /* sign-magnitude */

static void mp_zero(mp* a) { memset(a, 0, sizeof *a); }
static int mp_cmp_mag(const mp* a, const mp* b) {
  for (int i = MPN - 1; i >= 0; i--)
    if (a->w[i] != b->w[i]) return a->w[i] < b->w[i] ? -1 : 1;
  return 0;
}
static void mp_add_mag(mp* r, const mp* a, const mp* b) {
  uint64_t c = 0;
  for (int i = 0; i < MPN; i++) {
    c += (uint64_t)a->w[i] + b->w[i];
    r->w[i] = (uint32_t)c;
    c >>= 32;
  }
}
static void mp_sub_mag(mp* r, const mp* a, const mp* b) { /* |a| >= |b| */
  int64_t c = 0;
  for (int i = 0; i < MPN; i++) {
    c += (int64_t)a->w[i] - b->w[i];
    r->w[i] = (uint32_t)c;
    c >>= 32;
  }
}
static void mp_add(mp* r, const mp* a, const mp* b) {
  mp t;
  if (a->neg == b->neg) {
    mp_add_mag(&t, a, b);
    t.neg = a->neg;
  } else if (mp_cmp_mag(a, b) >= 0) {
    mp_sub_mag(&t, a, b);
    t.neg = a->neg;
  } else {
    mp_sub_mag(&t, b, a);
    t.neg = b->neg;
  }
  int z = 1;
  for (int i = 0; i < MPN; i++) z &= t.w[i] == 0;
  if (z) t.neg = 0;
  *r = t;
}
static void mp_negate(mp* a) {
  int z = 1;
  for (int i = 0; i < MPN; i++) z &= a->w[i] == 0;
  if (!z) a->neg = !a->neg;
}
static void mp_mul(mp* r, const mp* a, const mp* b) {
  mp t;
  mp_zero(&t);
  for (int i = 0; i < MPN; i++) {
    uint64_t c = 0;
    for (int j = 0; i + j < MPN; j++) {
      c += (uint64_t)a->w[i] * b->w[j] + t.w[i + j];
      t.w[i + j] = (uint32_t)c;
      c >>= 32;
    }
  }
  t.neg = a->neg != b->neg;
  int z = 1;
  for (int i = 0; i < MPN; i++) z &= t.w[i] == 0;
  if (z) t.neg = 0;
  *r = t;
}
/* truncating division of magnitudes: q = |a| / |d| (bitwise long division) */
static void mp_div_mag(mp* q, const mp* a, const mp* d) {
  mp rem, quo;
  mp_zero(&rem);
  mp_zero(&quo);
  for (int bit = MPN * 32 - 1; bit >= 0; bit--) {
    for (int i = MPN - 1; i > 0; i--) rem.w[i] = (rem.w[i] << 1) | (rem.w[i - 1] >> 31);
    rem.w[0] = (rem.w[0] << 1) | ((a->w[bit >> 5] >> (bit & 31)) & 1);
    if (mp_cmp_mag(&rem, d) >= 0) {
      mp_sub_mag(&rem, &rem, d);
      quo.w[bit >> 5] |= 1u << (bit & 31);
    }
  }
  *q = quo;
}
/* weierstrass.ts:106 divNearest(num, den) = (num + (num >= 0 ? den : -den) / 2n) / den, den > 0,
 * BigInt `/` truncating toward zero */
static void mp_div_nearest(mp* r, const mp* num, const mp* den) {
  mp half;
  mp one_;
  mp_zero(&one_);
  one_.w[0] = 2;
  mp_div_mag(&half, den, &one_);
  half.neg = num->neg; /* (+den)/2 or (-den)/2 truncated */
  mp t;
  mp_add(&t, num, &half);
  mp q;
  mp_div_mag(&q, &t, den);
  q.neg = t.neg;
  int z = 1;
  for (int i = 0; i < MPN; i++) z &= q.w[i] == 0;
  if (z) q.neg = 0;
  *r = q;
}
static void mp_from_hex(mp* a, const char* hex) {
  mp_zero(a);
  int n = (int)strlen(hex);
  for (int i = 0; i < n; i++) {
    char ch = hex[n - 1 - i];
    uint32_t v = ch <= '9' ? ch - '0' : (ch | 32) - 'a' + 10;
    a->w[i >> 3] |= v << (4 * (i & 7));
  }
}

/* ---- constants ---- */
static k1_ctx K1;
static k1_fe K1_BETA;
static mp GLV_A1, GLV_B1, GLV_A2, GLV_B2, K1_N;
static int inited = 0;

static void fe_set_hex(uint64_t* v, int nl, const char* hex) {
  memset(v, 0, nl * 8);
  int n = (int)strlen(hex);
  for (int i = 0; i < n; i++) {
    char ch = hex[n - 1 - i];
    uint64_t d = ch <= '9' ? ch - '0' : (ch | 32) - 'a' + 10;
    v[i >> 4] |= d << (4 * (i & 15));
  }
}
/* r = 2^k mod p by repeated modular doubling */
static void pow2_mod(uint64_t* r, const uint64_t* p, int nl, int k) {
  uint64_t t[8] = {1};
  for (int s = 0; s < k; s++) {
    uint64_t cy = 0;
    for (int i = 0; i < nl; i++) {
      uint64_t n = (t[i] << 1) | cy;
      cy = t[i] >> 63;
      t[i] = n;
    }
    /* if (cy || t >= p) t -= p */
    int ge = cy != 0;
    if (!ge) {
      ge = 1;
      for (int i = nl - 1; i >= 0; i--) {
        if (t[i] != p[i]) {
          ge = t[i] > p[i];
          break;
        }
      }
    }
    if (ge) {
      unsigned __int128 bw = 0;
      for (int i = 0; i < nl; i++) {
        unsigned __int128 d = (unsigned __int128)t[i] - p[i] - (uint64_t)bw;
        t[i] = (uint64_t)d;
        bw = (d >> 64) & 1;
      }
    }
  }
  memcpy(r, t, nl * 8);
}
static uint64_t neg_inv64(uint64_t p0) {
  uint64_t x = 1;
  for (int i = 0; i < 6; i++) x *= 2 - p0 * x;
  return (uint64_t)0 - x;
}

static void init_once(void) {
  if (inited) return;
  /* src/secp256k1.ts:48-56 */
  fe_set_hex(K1.p.v, 4, "fffffffffffffffffffffffffffffffffffffffffffffffffffffffefffffc2f");
  pow2_mod(K1.r1.v, K1.p.v, 4, 256);
  pow2_mod(K1.r2.v, K1.p.v, 4, 512);
  K1.inv = neg_inv64(K1.p.v[0]);
  k1_fe t;
  memset(&t, 0, sizeof t);
  t.v[0] = 21; /* b3 = 3*7, weierstrass.ts:612 */
  k1_tomont(&K1, &K1.b3, &t);
  /* src/secp256k1.ts:58-64 */
  fe_set_hex(t.v, 4, "7ae96a2b657c07106e64479eac3434e99cf0497512f58995c1396c28719501ee");
  k1_tomont(&K1, &K1_BETA, &t);
  mp_from_hex(&GLV_A1, "3086d221a7d46bcde86c90e49284eb15");
  mp_from_hex(&GLV_B1, "e4437ed6010e88286f547fa90abfe4c3");
  GLV_B1.neg = 1;
  mp_from_hex(&GLV_A2, "114ca50f7a8e2f3f657c1108d9d44cfd8");
  mp_from_hex(&GLV_B2, "3086d221a7d46bcde86c90e49284eb15");
  mp_from_hex(&K1_N, "fffffffffffffffffffffffffffffffebaaedce6af48a03bbfd25e8cd0364141");
  /* src/bls12-381.ts:134-148 */
  fe_set_hex(BLS.p.v, 6,
             "1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab");
  pow2_mod(BLS.r1.v, BLS.p.v, 6, 384);
  pow2_mod(BLS.r2.v, BLS.p.v, 6, 768);
  BLS.inv = neg_inv64(BLS.p.v[0]);
  bls_fe u;
  memset(&u, 0, sizeof u);
  u.v[0] = 12; /* b3 = 3*4 */
  bls_tomont(&BLS, &BLS.b3, &u);
  /* src/bls12-381.ts:321-345: G2 b = 4 + 4u, b3 = 12 + 12u; Montgomery one = (R mod p, 0) */
  memset(&G2C, 0, sizeof G2C);
  G2C.r1.c0 = BLS.r1;
  G2C.b3.c0 = BLS.b3;
  G2C.b3.c1 = BLS.b3;
  inited = 1;
}

/* weierstrass.ts:121-148 _splitEndoScalar; outputs |k1|, |k2| as 32 LE bytes each + signs */
static void split_endo(const uint8_t* k32, uint8_t* k1b, int* k1neg, uint8_t* k2b, int* k2neg) {
  mp k;
  mp_zero(&k);
  for (int i = 0; i < 32; i++) k.w[i >> 2] |= (uint32_t)k32[i] << (8 * (i & 3));
  mp t, c1, c2, k1, k2, u;
  mp_mul(&t, &GLV_B2, &k);
  mp_div_nearest(&c1, &t, &K1_N);
  mp nb1 = GLV_B1;
  mp_negate(&nb1);
  mp_mul(&t, &nb1, &k);
  mp_div_nearest(&c2, &t, &K1_N);
  /* k1 = k - c1*a1 - c2*a2 ; k2 = -c1*b1 - c2*b2 */
  mp_mul(&t, &c1, &GLV_A1);
  mp_negate(&t);
  mp_add(&k1, &k, &t);
  mp_mul(&t, &c2, &GLV_A2);
  mp_negate(&t);
  mp_add(&k1, &k1, &t);
  mp_mul(&t, &c1, &GLV_B1);
  mp_negate(&t);
  k2 = t;
  mp_mul(&u, &c2, &GLV_B2);
  mp_negate(&u);
  mp_add(&k2, &k2, &u);
  *k1neg = k1.neg;
  *k2neg = k2.neg;
  for (int i = 0; i < 32; i++) {
    k1b[i] = (uint8_t)(k1.w[i >> 2] >> (8 * (i & 3)));
    k2b[i] = (uint8_t)(k2.w[i >> 2] >> (8 * (i & 3)));
  }
}

static int is_zero_bytes(const uint8_t* b, int n) {
  uint8_t o = 0;
  for (int i = 0; i < n; i++) o |= b[i];
  return o == 0;
}

/* ---- exported ---- */
/* secp256k1 Point.multiplyUnsafe for a batch (weierstrass.ts:915-928) */
int orc_secp256k1_multiply_unsafe(const uint8_t* pts, const uint8_t* scalars, uint8_t* out, uint8_t* out_inf,
                                  size_t n) {
  init_once();
  for (size_t i = 0; i < n; i++) {
    k1_pt p, r;
    k1_pt_from_wire(&K1, &p, pts + 64 * i);
    const uint8_t* k = scalars + 32 * i;
    int one = k[0] == 1 && is_zero_bytes(k + 1, 31);
    if (is_zero_bytes(k, 32) || k1_is0(&p.Z)) {
      k1_pt_zero(&K1, &r);                         /* :921 */
    } else if (one) {
      r = p;                                       /* :922 */
    } else {
      uint8_t sc[2 * 32];
      int n1, n2;
      split_endo(k, sc, &n1, sc + 32, &n2);        /* pushWnafPair :660-671 */
      k1_pt pp[2];
      pp[0] = p;
      pp[1] = p;
      k1_mul(&K1, &pp[1].X, &p.X, &K1_BETA);       /* psi(P) = (beta*X, Y, Z) :664 */
      if (n1) k1_pt_neg(&K1, &pp[0], &pp[0]);
      if (n2) k1_pt_neg(&K1, &pp[1], &pp[1]);
      k1_mul_add_unsafe(&K1, &r, pp, sc, 32, 2);
    }
    out_inf[i] = (uint8_t)k1_pt_to_wire(&K1, out + 64 * i, &r);
  }
  return 0;
}

/* bls12-381 G1 pippenger (curve.ts:863-905), Fn.BITS = 255 */
int orc_bls12_381_g1_pippenger(const uint8_t* pts, const uint8_t* scalars, size_t n, uint8_t* out, uint8_t* out_inf) {
  init_once();
  bls_pt* P = (bls_pt*)malloc((n ? n : 1) * sizeof(bls_pt));
  for (size_t i = 0; i < n; i++) bls_pt_from_wire(&BLS, &P[i], pts + 96 * i);
  bls_pt r;
  bls_pippenger(&BLS, &r, P, scalars, n, 255);
  free(P);
  *out_inf = (uint8_t)bls_pt_to_wire(&BLS, out, &r);
  return 0;
}

/* secp256k1 pippenger, Fn.BITS = 256 (test/point.test.ts runs MSM on every curve) */
int orc_secp256k1_pippenger(const uint8_t* pts, const uint8_t* scalars, size_t n, uint8_t* out, uint8_t* out_inf) {
  init_once();
  k1_pt* P = (k1_pt*)malloc((n ? n : 1) * sizeof(k1_pt));
  for (size_t i = 0; i < n; i++) k1_pt_from_wire(&K1, &P[i], pts + 64 * i);
  k1_pt r;
  k1_pippenger(&K1, &r, P, scalars, n, 256);
  free(P);
  *out_inf = (uint8_t)k1_pt_to_wire(&K1, out, &r);
  return 0;
}

/* bls12-381 G1 Point.multiplyUnsafe batch (no endomorphism: single wNAF-4 stream) */
int orc_bls12_381_g1_multiply_unsafe(const uint8_t* pts, const uint8_t* scalars, uint8_t* out, uint8_t* out_inf,
                                     size_t n) {
  init_once();
  for (size_t i = 0; i < n; i++) {
    bls_pt p, r;
    bls_pt_from_wire(&BLS, &p, pts + 96 * i);
    const uint8_t* k = scalars + 32 * i;
    if (is_zero_bytes(k, 32) || bls_is0(&p.Z)) bls_pt_zero(&BLS, &r);
    else bls_mul_add_unsafe(&BLS, &r, &p, k, 32, 1);
    out_inf[i] = (uint8_t)bls_pt_to_wire(&BLS, out + 96 * i, &r);
  }
  return 0;
}

/* bls12-381 G2 pippenger (curve.ts:863-905 over Fp2 points), Fn.BITS = 255; wire 192 bytes per point */
int orc_bls12_381_g2_pippenger(const uint8_t* pts, const uint8_t* scalars, size_t n, uint8_t* out, uint8_t* out_inf) {
  init_once();
  g2_pt* P = (g2_pt*)malloc((n ? n : 1) * sizeof(g2_pt));
  for (size_t i = 0; i < n; i++) g2_pt_from_wire(&G2C, &P[i], pts + 192 * i);
  g2_pt r;
  g2_pippenger(&G2C, &r, P, scalars, n, 255);
  free(P);
  *out_inf = (uint8_t)g2_pt_to_wire(&G2C, out, &r);
  return 0;
}

/* secp256k1 Point.multiply for a batch (weierstrass.ts:900-907 -> mulSecret curve.ts:741-750 -> mulCTBlinded
 * :663-690 / mulCT :658-661 -> runCT :647-656 -> fixedWindowCT :707-729, the path of an un-precomputed point:
 * BASELINE configs[0], benchmark/point.ts:31).  blinds: n x 16 bytes (the RNG output, big-endian as
 * bytesToNumberBE reads it; top two bits forced to 10 as :683), or NULL for the unblinded mulCT shape.
 * Scalars must satisfy 1 <= k < n (the caller's contract; weierstrass.ts:904). */
int orc_secp256k1_multiply(const uint8_t* pts, const uint8_t* scalars, const uint8_t* blinds, uint8_t* out, uint8_t* out_inf,
                           size_t n) {
  init_once();
  enum { W = 5, SIZE = 1 << W };
  for (size_t i = 0; i < n; i++) {
    k1_pt p, table[SIZE], acc, sel;
    k1_pt_from_wire(&K1, &p, pts + 64 * i);
    mp nn;
    mp_zero(&nn);
    for (int j = 0; j < 32; j++) nn.w[j >> 2] |= (uint32_t)scalars[32 * i + j] << (8 * (j & 3));
    int bits = 256;
    if (blinds) { /* n = scalar + blind * Fn.ORDER, bits = Fn.BITS + 128 (:670, :689) */
      mp b, prod;
      mp_zero(&b);
      uint8_t bb[16];
      memcpy(bb, blinds + 16 * i, 16);
      bb[0] = (uint8_t)((bb[0] & 0x3f) | 0x80);
      for (int j = 0; j < 16; j++) b.w[j >> 2] |= (uint32_t)bb[15 - j] << (8 * (j & 3));
      mp_mul(&prod, &b, &K1_N);
      mp_add(&nn, &nn, &prod);
      bits = 256 + 128;
    }
    k1_pt_zero(&K1, &table[0]); /* flat table [O, P, 2P, ..., 31P] */
    for (int t = 1; t < SIZE; t++) k1_pt_add(&K1, &table[t], &table[t - 1], &p);
    const int windows = (bits + W - 1) / W;
    k1_pt_zero(&K1, &acc);
    for (int w = windows - 1; w >= 0; w--) {
      if (w != windows - 1)
        for (int d = 0; d < W; d++) k1_pt_double(&K1, &acc, &acc);
      const int bit = w * W;
      uint32_t digit = nn.w[bit >> 5] >> (bit & 31);
      if ((bit & 31) > 32 - W && (bit >> 5) + 1 < MPN) digit |= nn.w[(bit >> 5) + 1] << (32 - (bit & 31));
      digit &= SIZE - 1;
      sel = table[0]; /* data-oblivious scan over every entry (:722-723) */
      for (int t = 1; t < SIZE; t++)
        if ((uint32_t)t == digit) sel = table[t];
      k1_pt_add(&K1, &acc, &acc, &sel); /* one add per window, even for digit 0 */
    }
    out_inf[i] = (uint8_t)k1_pt_to_wire(&K1, out + 64 * i, &acc);
  }
  return 0;
}

/* ======================================================================================
 * ed25519 verification (src/abstract/edwards.ts:942-989) - restated for CPU timing.
 * Field: the NL = 4 Montgomery template over p = 2^255 - 19 (only its field functions are used);
 * group law: dbl-2008-hwcd / add-2008-hwcd exactly as edwards.ts:505-545 (a = -1).
 * ====================================================================================== */
#define NL 4
#define PFX(x) ed_##x
#include "field_tmpl.h"
#undef NL
#undef PFX

typedef struct { ed_fe X, Y, Z, T; } ed_ext;
static ed_ctx ED;
static ed_fe ED_D, ED_SQRT_M1, ED_ONE;
static ed_ext ED_BASE;
static uint8_t ED_L_BYTES[32];
static int ed_inited = 0;

static void ed_init_once(void) {
  if (ed_inited) return;
  fe_set_hex(ED.p.v, 4, "7fffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffed");
  pow2_mod(ED.r1.v, ED.p.v, 4, 256);
  pow2_mod(ED.r2.v, ED.p.v, 4, 512);
  ED.inv = neg_inv64(ED.p.v[0]);
  ED_ONE = ED.r1;
  ed_fe t;
  fe_set_hex(t.v, 4, "52036cee2b6ffe738cc740797779e89800700a4d4141d8ab75eb4dca135978a3"); /* ed25519.ts:61 */
  ed_tomont(&ED, &ED_D, &t);
  fe_set_hex(t.v, 4, "2b8324804fc1df0b2b4d00993dfbd7a72f431806ad2fe478c4ee1b274a0ea0b0"); /* ed25519.ts:102-104 */
  ed_tomont(&ED, &ED_SQRT_M1, &t);
  fe_set_hex(t.v, 4, "216936d3cd6e53fec0a4e231fdd6dc5c692cc7609525a7b2c9562d608f25d51a");
  ed_tomont(&ED, &ED_BASE.X, &t);
  fe_set_hex(t.v, 4, "6666666666666666666666666666666666666666666666666666666666666658");
  ed_tomont(&ED, &ED_BASE.Y, &t);
  ED_BASE.Z = ED_ONE;
  ed_mul(&ED, &ED_BASE.T, &ED_BASE.X, &ED_BASE.Y);
  uint64_t l[4];
  fe_set_hex(l, 4, "1000000000000000000000000000000014def9dea2f79cd65812631a5cf5d3ed");
  memcpy(ED_L_BYTES, l, 32);
  ed_inited = 1;
}

static void ed_ext_zero(ed_ext* r) { /* (0, 1, 1, 0) edwards.ts:370 */
  memset(r, 0, sizeof *r);
  r->Y = ED_ONE;
  r->Z = ED_ONE;
}
static void ed_ext_double(ed_ext* r, const ed_ext* p) { /* edwards.ts:505-521 */
  ed_fe A, B, C, D, E, G, F, H, t;
  ed_mul(&ED, &A, &p->X, &p->X);
  ed_mul(&ED, &B, &p->Y, &p->Y);
  ed_mul(&ED, &C, &p->Z, &p->Z);
  ed_add(&ED, &C, &C, &C);
  ed_neg(&ED, &D, &A); /* a = -1 */
  ed_add(&ED, &t, &p->X, &p->Y);
  ed_mul(&ED, &E, &t, &t);
  ed_sub(&ED, &E, &E, &A);
  ed_sub(&ED, &E, &E, &B);
  ed_add(&ED, &G, &D, &B);
  ed_sub(&ED, &F, &G, &C);
  ed_sub(&ED, &H, &D, &B);
  ed_mul(&ED, &r->X, &E, &F);
  ed_mul(&ED, &r->Y, &G, &H);
  ed_mul(&ED, &r->T, &E, &H);
  ed_mul(&ED, &r->Z, &F, &G);
}
static void ed_ext_add(ed_ext* r, const ed_ext* p, const ed_ext* q) { /* edwards.ts:526-545 */
  ed_fe A, B, C, D, E, F, G, H, t, u;
  ed_mul(&ED, &A, &p->X, &q->X);
  ed_mul(&ED, &B, &p->Y, &q->Y);
  ed_mul(&ED, &C, &p->T, &ED_D);
  ed_mul(&ED, &C, &C, &q->T);
  ed_mul(&ED, &D, &p->Z, &q->Z);
  ed_add(&ED, &t, &p->X, &p->Y);
  ed_add(&ED, &u, &q->X, &q->Y);
  ed_mul(&ED, &E, &t, &u);
  ed_sub(&ED, &E, &E, &A);
  ed_sub(&ED, &E, &E, &B);
  ed_sub(&ED, &F, &D, &C);
  ed_add(&ED, &G, &D, &C);
  ed_add(&ED, &H, &B, &A); /* B - a*A, a = -1 */
  ed_mul(&ED, &r->X, &E, &F);
  ed_mul(&ED, &r->Y, &G, &H);
  ed_mul(&ED, &r->T, &E, &H);
  ed_mul(&ED, &r->Z, &F, &G);
}
static void ed_ext_neg(ed_ext* r, const ed_ext* p) {
  *r = *p;
  ed_neg(&ED, &r->X, &p->X);
  ed_neg(&ED, &r->T, &p->T);
}
static int ed_ext_is0(const ed_ext* p) { /* equals(ZERO): X*1 == 0*Z and Y*1 == 1*Z */
  return ed_is0(&p->X) && ed_eq(&p->Y, &p->Z);
}
static void ed_pow2k(ed_fe* r, const ed_fe* x, int k) {
  *r = *x;
  for (int i = 0; i < k; i++) ed_mul(&ED, r, r, r);
}
/* ed25519.ts:67-86 */
static void ed_pow_p58(ed_fe* out, const ed_fe* x) {
  ed_fe x2, b2, b4, b5, b10, b20, b40, b80, b160, b240, b250, t;
  ed_mul(&ED, &x2, x, x);
  ed_mul(&ED, &b2, &x2, x);
  ed_pow2k(&t, &b2, 2); ed_mul(&ED, &b4, &t, &b2);
  ed_pow2k(&t, &b4, 1); ed_mul(&ED, &b5, &t, x);
  ed_pow2k(&t, &b5, 5); ed_mul(&ED, &b10, &t, &b5);
  ed_pow2k(&t, &b10, 10); ed_mul(&ED, &b20, &t, &b10);
  ed_pow2k(&t, &b20, 20); ed_mul(&ED, &b40, &t, &b20);
  ed_pow2k(&t, &b40, 40); ed_mul(&ED, &b80, &t, &b40);
  ed_pow2k(&t, &b80, 80); ed_mul(&ED, &b160, &t, &b80);
  ed_pow2k(&t, &b160, 80); ed_mul(&ED, &b240, &t, &b80);
  ed_pow2k(&t, &b240, 10); ed_mul(&ED, &b250, &t, &b10);
  ed_pow2k(&t, &b250, 2); ed_mul(&ED, out, &t, x);
}
/* ed25519.ts:107-125 */
static int ed_uv_ratio(ed_fe* xout, const ed_fe* u, const ed_fe* v) {
  ed_fe v3, v7, pw, x, vx2, root2, negu, t;
  ed_mul(&ED, &v3, v, v); ed_mul(&ED, &v3, &v3, v);
  ed_mul(&ED, &v7, &v3, &v3); ed_mul(&ED, &v7, &v7, v);
  ed_mul(&ED, &t, u, &v7);
  ed_pow_p58(&pw, &t);
  ed_mul(&ED, &x, u, &v3); ed_mul(&ED, &x, &x, &pw);
  ed_mul(&ED, &vx2, &x, &x); ed_mul(&ED, &vx2, &vx2, v);
  ed_mul(&ED, &root2, &x, &ED_SQRT_M1);
  ed_neg(&ED, &negu, u);
  int useRoot1 = ed_eq(&vx2, u), useRoot2 = ed_eq(&vx2, &negu);
  ed_mul(&ED, &t, &negu, &ED_SQRT_M1);
  int noRoot = ed_eq(&vx2, &t);
  if (useRoot2 || noRoot) x = root2;
  ed_fe xc;
  ed_frommont(&ED, &xc, &x);
  if (xc.v[0] & 1) ed_neg(&ED, &x, &x);
  *xout = x;
  return useRoot1 || useRoot2;
}
/* edwards.ts:405-436 */
static int ed_from_bytes(ed_ext* P, const uint8_t* b, int zip215) {
  uint8_t n[32];
  memcpy(n, b, 32);
  int sign = (n[31] & 0x80) != 0;
  n[31] &= 0x7f;
  ed_fe yraw, y, y2, u, v, x;
  memcpy(&yraw, n, 32);
  if (!zip215) { /* y < p */
    int lt = 0;
    for (int i = 3; i >= 0; i--) {
      if (yraw.v[i] != ED.p.v[i]) { lt = yraw.v[i] < ED.p.v[i]; break; }
    }
    if (!lt) return 0;
  }
  ed_tomont(&ED, &y, &yraw);
  ed_mul(&ED, &y2, &y, &y);
  ed_sub(&ED, &u, &y2, &ED_ONE);
  ed_mul(&ED, &v, &ED_D, &y2);
  ed_add(&ED, &v, &v, &ED_ONE);
  if (!ed_uv_ratio(&x, &u, &v)) return 0;
  int x0 = ed_is0(&x);
  if (!zip215 && x0 && sign) return 0;
  if (sign && !x0) ed_neg(&ED, &x, &x);
  P->X = x; P->Y = y; P->Z = ED_ONE;
  ed_mul(&ED, &P->T, &x, &y);
  return 1;
}
/* wNAF-4 walk (curve.ts:820-836 with one point), scalar as nbytes LE */
static void ed_mul_unsafe(ed_ext* out, const ed_ext* p, const uint8_t* k, int nbytes) {
  ed_ext table[4], dbl, acc, item;
  int8_t digits[600];
  ed_ext_double(&dbl, p);
  table[0] = *p;
  for (int j = 1; j < 4; j++) ed_ext_add(&table[j], &table[j - 1], &dbl);
  int len = ed_wnaf4(digits, k, nbytes);
  ed_ext_zero(&acc);
  for (int bit = len - 1; bit >= 0; bit--) {
    if (bit != len - 1) ed_ext_double(&acc, &acc);
    int w = digits[bit];
    if (w) {
      item = table[((w < 0 ? -w : w) - 1) >> 1];
      if (w < 0) ed_ext_neg(&item, &item);
      ed_ext_add(&acc, &acc, &item);
    }
  }
  *out = acc;
}
/* eddsa.verify (edwards.ts:942-989) with the challenge k already hashed (32 bytes LE, < L) */
int orc_ed25519_verify_batch(const uint8_t* sigs, const uint8_t* pks, const uint8_t* ks, int zip215, uint8_t* out,
                             size_t n) {
  init_once();
  ed_init_once();
  for (size_t i = 0; i < n; i++) {
    const uint8_t *sig = sigs + 64 * i, *pk = pks + 32 * i, *k = ks + 32 * i;
    ed_ext A, R, SB, kA, RkA, t;
    out[i] = 0;
    if (!ed_from_bytes(&A, pk, zip215)) continue;
    if (!ed_from_bytes(&R, sig, zip215)) continue;
    /* s < L (BASE.multiplyUnsafe throws otherwise, edwards.ts:573) */
    int lt = 0;
    for (int j = 31; j >= 0; j--) {
      if (sig[32 + j] != ED_L_BYTES[j]) { lt = sig[32 + j] < ED_L_BYTES[j]; break; }
    }
    if (!lt) continue;
    ed_mul_unsafe(&SB, &ED_BASE, sig + 32, 32);
    if (!zip215) { /* isSmallOrder */
      ed_ext_double(&t, &A); ed_ext_double(&t, &t); ed_ext_double(&t, &t);
      if (ed_ext_is0(&t)) continue;
    }
    if (is_zero_bytes(k, 32)) ed_ext_zero(&kA);
    else ed_mul_unsafe(&kA, &A, k, 32);
    ed_ext_add(&RkA, &R, &kA);
    ed_ext_neg(&t, &SB);
    ed_ext_add(&t, &RkA, &t);
    ed_ext_double(&t, &t); ed_ext_double(&t, &t); ed_ext_double(&t, &t);
    out[i] = (uint8_t)ed_ext_is0(&t);
  }
  return 0;
}

/* ======================================================================================
 * FFT over the bls12-381 scalar field Fr: FFT(roots, Fr).direct / .inverse
 * (src/abstract/fft.ts:518-577) through the reference's own loop shapes: natural in / natural out
 * is bitReversalInplace + DIT butterflies (FFTCore :445-480 with dit = true, brp = true), the
 * brpOutput form is DIF without the final permutation.  Roots table: rootsOfUnity.roots(bits)
 * (:262-277) built from omega by the multiplication chain; inverse walks the reversed table
 * (:296-304) and scales by 1/N (:566-571).  Values: canonical 32-byte LE residues.
 * ====================================================================================== */
#define NL 4
#define PFX(x) fr_##x
#include "field_tmpl.h"
#undef NL
#undef PFX

static fr_ctx FR;
static int fr_inited = 0;
static void fr_init_once(void) {
  if (fr_inited) return;
  fe_set_hex(FR.p.v, 4, "73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001"); /* bls12-381.ts Fr */
  pow2_mod(FR.r1.v, FR.p.v, 4, 256);
  pow2_mod(FR.r2.v, FR.p.v, 4, 512);
  FR.inv = neg_inv64(FR.p.v[0]);
  fr_inited = 1;
}
static size_t brev(size_t i, int bits) {
  size_t r = 0;
  for (int b = 0; b < bits; b++) r |= ((i >> b) & 1) << (bits - 1 - b);
  return r;
}
/* flags: bit 0 inverse, bit 1 brpInput, bit 2 brpOutput.  omega32: primitive 2^bits-th root. */
int orc_fft_fr(int bits, const uint8_t* omega32, const uint8_t* in, uint8_t* out, int flags) {
  fr_init_once();
  const size_t N = (size_t)1 << bits;
  const int inverse = flags & 1, brp_in = (flags >> 1) & 1, brp_out = (flags >> 2) & 1;
  fr_fe* v = (fr_fe*)malloc(N * sizeof(fr_fe));
  fr_fe* roots = (fr_fe*)malloc(N * sizeof(fr_fe));
  if (!v || !roots) return -1;
  fr_fe w, t;
  memcpy(t.v, omega32, 32);
  fr_tomont(&FR, &w, &t);
  roots[0] = FR.r1;
  for (size_t k = 1; k < N; k++) fr_mul(&FR, &roots[k], &roots[k - 1], &w);
  for (size_t i = 0; i < N; i++) {
    memcpy(t.v, in + 32 * i, 32);
    fr_tomont(&FR, &v[i], &t);
  }
  const int dit = !(brp_out && !brp_in) && !(brp_in && brp_out); /* FFT.getLoop :530-541 */
  if (brp_in && brp_out && bits) /* core(bitReversalInplace(values)) with DIF, brp = false */
    for (size_t i = 0; i < N; i++) {
      size_t j = brev(i, bits);
      if (i < j) { fr_fe s = v[i]; v[i] = v[j]; v[j] = s; }
    }
  if (dit && !brp_in && bits) /* dit && brp: bitReversalInplace first */
    for (size_t i = 0; i < N; i++) {
      size_t j = brev(i, bits);
      if (i < j) { fr_fe s = v[i]; v[i] = v[j]; v[j] = s; }
    }
  for (int i = 0; i < bits; i++) {
    const int s = dit ? i + 1 : bits - i;
    const size_t m = (size_t)1 << s, m2 = m >> 1, stride = N >> s;
    for (size_t k = 0; k < N; k += m)
      for (size_t j = 0; j < m2; j++) {
        size_t pos = j * stride;
        if (inverse && pos) pos = N - pos; /* roots.inverse(bits)[pos] */
        fr_fe a = v[k + j], b = v[k + j + m2], x;
        if (dit) {
          fr_mul(&FR, &x, &b, &roots[pos]);
          fr_add(&FR, &v[k + j], &a, &x);
          fr_sub(&FR, &v[k + j + m2], &a, &x);
        } else {
          fr_add(&FR, &v[k + j], &a, &b);
          fr_sub(&FR, &x, &a, &b);
          fr_mul(&FR, &v[k + j + m2], &x, &roots[pos]);
        }
      }
  }
  fr_fe ninv = FR.r1;
  if (inverse) { /* 1/N = ((r + 1) / 2)^bits */
    fr_fe half, one;
    memset(&one, 0, sizeof one);
    one.v[0] = 1;
    /* (r + 1) / 2 */
    unsigned __int128 cy = 1;
    uint64_t tmp[4];
    for (int i = 0; i < 4; i++) { cy += FR.p.v[i]; tmp[i] = (uint64_t)cy; cy >>= 64; }
    for (int i = 0; i < 4; i++) t.v[i] = (tmp[i] >> 1) | (i < 3 ? tmp[i + 1] << 63 : (uint64_t)cy << 63);
    fr_tomont(&FR, &half, &t);
    for (int i = 0; i < bits; i++) fr_mul(&FR, &ninv, &ninv, &half);
  }
  fr_fe one_raw;
  memset(&one_raw, 0, sizeof one_raw);
  one_raw.v[0] = 1;
  for (size_t i = 0; i < N; i++) {
    fr_fe x = v[i];
    if (inverse) fr_mul(&FR, &x, &x, &ninv);
    fr_mul(&FR, &t, &x, &one_raw); /* out of Montgomery form */
    memcpy(out + 32 * i, t.v, 32);
  }
  free(v);
  free(roots);
  return 0;
}
