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
static bls_ctx BLS;
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
