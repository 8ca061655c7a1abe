/* Field + short-Weierstrass template, instantiated per limb count (NL = 4 or 6 64-bit limbs).
 * TEST INFRASTRUCTURE (oracle) - see ../__init__.py.
 *
 * Montgomery arithmetic replaces the reference's BigInt `a*b % p` (src/abstract/modular.ts:50-54,
 * :956) - values are identical after conversion; the group law follows the reference line by
 * line: Renes-Costello-Batina complete projective formulas, src/abstract/weierstrass.ts:793-828
 * (double) and :834-880 (add), with a = 0 so mulA() returns 0 (:613) and b3 = 3*b (:612).
 *
 * Instantiate with:  #define NL 6 / #define PFX(x) bls_##x / #include "field_tmpl.h"
 * With PT_ONLY defined only the group-law part is instantiated, over a field the includer supplies under the
 * same names (PFX(fe) of NL * 8 wire bytes, PFX(ctx) with r1 and b3, is0 / add / sub / mul / neg / inv /
 * tomont / frommont): bls12-381 G2 over Fp2 (src/abstract/tower.ts:393-475), see oracle.c.
 */
#ifndef PT_ONLY
typedef struct { uint64_t v[NL]; } PFX(fe);
typedef struct {
  PFX(fe) p, r1, r2, b3; /* modulus, R mod p, R^2 mod p, 3b (Montgomery form) */
  uint64_t inv;          /* -p^-1 mod 2^64 */
} PFX(ctx);

static inline int PFX(is0)(const PFX(fe) * a) {
  uint64_t o = 0;
  for (int i = 0; i < NL; i++) o |= a->v[i];
  return o == 0;
}
static inline int PFX(eq)(const PFX(fe) * a, const PFX(fe) * b) {
  uint64_t o = 0;
  for (int i = 0; i < NL; i++) o |= a->v[i] ^ b->v[i];
  return o == 0;
}
static inline void PFX(csub)(const PFX(ctx) * c, PFX(fe) * a, uint64_t hi) {
  PFX(fe) s;
  unsigned __int128 bw = 0;
  for (int i = 0; i < NL; i++) {
    unsigned __int128 t = (unsigned __int128)a->v[i] - c->p.v[i] - (uint64_t)bw;
    s.v[i] = (uint64_t)t;
    bw = (t >> 64) & 1;
  }
  if (hi || !bw) *a = s;
}
static inline void PFX(add)(const PFX(ctx) * c, PFX(fe) * r, const PFX(fe) * a, const PFX(fe) * b) {
  unsigned __int128 cy = 0;
  for (int i = 0; i < NL; i++) {
    cy += (unsigned __int128)a->v[i] + b->v[i];
    r->v[i] = (uint64_t)cy;
    cy >>= 64;
  }
  PFX(csub)(c, r, (uint64_t)cy);
}
static inline void PFX(sub)(const PFX(ctx) * c, PFX(fe) * r, const PFX(fe) * a, const PFX(fe) * b) {
  unsigned __int128 bw = 0;
  for (int i = 0; i < NL; i++) {
    unsigned __int128 t = (unsigned __int128)a->v[i] - b->v[i] - (uint64_t)bw;
    r->v[i] = (uint64_t)t;
    bw = (t >> 64) & 1;
  }
  if (bw) {
    unsigned __int128 cy = 0;
    for (int i = 0; i < NL; i++) {
      cy += (unsigned __int128)r->v[i] + c->p.v[i];
      r->v[i] = (uint64_t)cy;
      cy >>= 64;
    }
  }
}
static inline void PFX(mul)(const PFX(ctx) * c, PFX(fe) * r, const PFX(fe) * a, const PFX(fe) * b) {
  uint64_t T[NL + 2];
  for (int i = 0; i < NL + 2; i++) T[i] = 0;
  for (int i = 0; i < NL; i++) {
    unsigned __int128 cy = 0;
    for (int j = 0; j < NL; j++) {
      cy += (unsigned __int128)a->v[j] * b->v[i] + T[j];
      T[j] = (uint64_t)cy;
      cy >>= 64;
    }
    cy += T[NL];
    T[NL] = (uint64_t)cy;
    T[NL + 1] = (uint64_t)(cy >> 64);
    uint64_t m = T[0] * c->inv;
    cy = ((unsigned __int128)m * c->p.v[0] + T[0]) >> 64;
    for (int j = 1; j < NL; j++) {
      cy += (unsigned __int128)m * c->p.v[j] + T[j];
      T[j - 1] = (uint64_t)cy;
      cy >>= 64;
    }
    cy += T[NL];
    T[NL - 1] = (uint64_t)cy;
    T[NL] = T[NL + 1] + (uint64_t)(cy >> 64);
  }
  for (int i = 0; i < NL; i++) r->v[i] = T[i];
  PFX(csub)(c, r, T[NL]);
}
static inline void PFX(neg)(const PFX(ctx) * c, PFX(fe) * r, const PFX(fe) * a) {
  PFX(fe) z;
  memset(&z, 0, sizeof z);
  PFX(sub)(c, r, &z, a);
}
static void PFX(tomont)(const PFX(ctx) * c, PFX(fe) * r, const PFX(fe) * a) { PFX(mul)(c, r, a, &c->r2); }
static void PFX(frommont)(const PFX(ctx) * c, PFX(fe) * r, const PFX(fe) * a) {
  PFX(fe) one;
  memset(&one, 0, sizeof one);
  one.v[0] = 1;
  PFX(mul)(c, r, a, &one);
}
/* a^(p-2): same value as the reference's Euclidean invert (modular.ts:159-182) */
static void PFX(inv)(const PFX(ctx) * c, PFX(fe) * r, const PFX(fe) * a) {
  PFX(fe) e = c->p, acc = c->r1, base = *a;
  e.v[0] -= 2;
  for (int i = 0; i < NL * 64; i++) {
    if ((e.v[i / 64] >> (i % 64)) & 1) PFX(mul)(c, &acc, &acc, &base);
    PFX(mul)(c, &base, &base, &base);
  }
  *r = acc;
}
#endif /* PT_ONLY */

typedef struct { PFX(fe) X, Y, Z; } PFX(pt);

/* ZERO = (0, 1, 0), weierstrass.ts:687 */
static void PFX(pt_zero)(const PFX(ctx) * c, PFX(pt) * r) {
  memset(r, 0, sizeof *r);
  r->Y = c->r1;
}
/* weierstrass.ts:793-828, a = 0 */
static void PFX(pt_double)(const PFX(ctx) * c, PFX(pt) * r, const PFX(pt) * p) {
  PFX(fe) t0, t1, t2, t3, X3, Y3, Z3;
  const PFX(fe) *X1 = &p->X, *Y1 = &p->Y, *Z1 = &p->Z;
  PFX(mul)(c, &t0, X1, X1);
  PFX(mul)(c, &t1, Y1, Y1);
  PFX(mul)(c, &t2, Z1, Z1);
  PFX(mul)(c, &t3, X1, Y1);
  PFX(add)(c, &t3, &t3, &t3);
  PFX(mul)(c, &Z3, X1, Z1);
  PFX(add)(c, &Z3, &Z3, &Z3);
  memset(&X3, 0, sizeof X3);               /* X3 = mulA(Z3) = 0 */
  PFX(mul)(c, &Y3, &c->b3, &t2);
  PFX(add)(c, &Y3, &X3, &Y3);
  PFX(sub)(c, &X3, &t1, &Y3);
  PFX(add)(c, &Y3, &t1, &Y3);
  PFX(mul)(c, &Y3, &X3, &Y3);
  PFX(mul)(c, &X3, &t3, &X3);
  PFX(mul)(c, &Z3, &c->b3, &Z3);
  memset(&t2, 0, sizeof t2);               /* t2 = mulA(t2) = 0 */
  PFX(sub)(c, &t3, &t0, &t2);
  memset(&t3, 0, sizeof t3);               /* t3 = mulA(t3) = 0 */
  PFX(add)(c, &t3, &t3, &Z3);
  PFX(add)(c, &Z3, &t0, &t0);
  PFX(add)(c, &t0, &Z3, &t0);
  PFX(add)(c, &t0, &t0, &t2);
  PFX(mul)(c, &t0, &t0, &t3);
  PFX(add)(c, &Y3, &Y3, &t0);
  PFX(mul)(c, &t2, Y1, Z1);
  PFX(add)(c, &t2, &t2, &t2);
  PFX(mul)(c, &t0, &t2, &t3);
  PFX(sub)(c, &X3, &X3, &t0);
  PFX(mul)(c, &Z3, &t2, &t1);
  PFX(add)(c, &Z3, &Z3, &Z3);
  PFX(add)(c, &Z3, &Z3, &Z3);
  r->X = X3;
  r->Y = Y3;
  r->Z = Z3;
}
/* weierstrass.ts:834-880, a = 0 */
static void PFX(pt_add)(const PFX(ctx) * c, PFX(pt) * r, const PFX(pt) * p, const PFX(pt) * q) {
  PFX(fe) t0, t1, t2, t3, t4, t5, X3, Y3, Z3;
  const PFX(fe) *X1 = &p->X, *Y1 = &p->Y, *Z1 = &p->Z, *X2 = &q->X, *Y2 = &q->Y, *Z2 = &q->Z;
  PFX(mul)(c, &t0, X1, X2);
  PFX(mul)(c, &t1, Y1, Y2);
  PFX(mul)(c, &t2, Z1, Z2);
  PFX(add)(c, &t3, X1, Y1);
  PFX(add)(c, &t4, X2, Y2);
  PFX(mul)(c, &t3, &t3, &t4);
  PFX(add)(c, &t4, &t0, &t1);
  PFX(sub)(c, &t3, &t3, &t4);
  PFX(add)(c, &t4, X1, Z1);
  PFX(add)(c, &t5, X2, Z2);
  PFX(mul)(c, &t4, &t4, &t5);
  PFX(add)(c, &t5, &t0, &t2);
  PFX(sub)(c, &t4, &t4, &t5);
  PFX(add)(c, &t5, Y1, Z1);
  PFX(add)(c, &X3, Y2, Z2);
  PFX(mul)(c, &t5, &t5, &X3);
  PFX(add)(c, &X3, &t1, &t2);
  PFX(sub)(c, &t5, &t5, &X3);
  memset(&Z3, 0, sizeof Z3);               /* Z3 = mulA(t4) = 0 */
  PFX(mul)(c, &X3, &c->b3, &t2);
  PFX(add)(c, &Z3, &X3, &Z3);
  PFX(sub)(c, &X3, &t1, &Z3);
  PFX(add)(c, &Z3, &t1, &Z3);
  PFX(mul)(c, &Y3, &X3, &Z3);
  PFX(add)(c, &t1, &t0, &t0);
  PFX(add)(c, &t1, &t1, &t0);
  memset(&t2, 0, sizeof t2);               /* t2 = mulA(t2) = 0 */
  PFX(mul)(c, &t4, &c->b3, &t4);
  PFX(add)(c, &t1, &t1, &t2);
  PFX(sub)(c, &t2, &t0, &t2);
  memset(&t2, 0, sizeof t2);               /* t2 = mulA(t2) = 0 */
  PFX(add)(c, &t4, &t4, &t2);
  PFX(mul)(c, &t0, &t1, &t4);
  PFX(add)(c, &Y3, &Y3, &t0);
  PFX(mul)(c, &t0, &t5, &t4);
  PFX(mul)(c, &X3, &t3, &X3);
  PFX(sub)(c, &X3, &X3, &t0);
  PFX(mul)(c, &t0, &t3, &t1);
  PFX(mul)(c, &Z3, &t5, &Z3);
  PFX(add)(c, &Z3, &Z3, &t0);
  r->X = X3;
  r->Y = Y3;
  r->Z = Z3;
}
static void PFX(pt_neg)(const PFX(ctx) * c, PFX(pt) * r, const PFX(pt) * p) {
  r->X = p->X;
  r->Z = p->Z;
  PFX(neg)(c, &r->Y, &p->Y);
}
/* weierstrass.ts:951-969: (X/Z, Y/Z); ZERO -> (0,0); returns 1 if infinity */
static int PFX(pt_to_affine)(const PFX(ctx) * c, PFX(fe) * x, PFX(fe) * y, const PFX(pt) * p) {
  if (PFX(is0)(&p->Z)) {
    memset(x, 0, sizeof *x);
    memset(y, 0, sizeof *y);
    return 1;
  }
  PFX(fe) iz;
  PFX(inv)(c, &iz, &p->Z);
  PFX(mul)(c, x, &p->X, &iz);
  PFX(mul)(c, y, &p->Y, &iz);
  return 0;
}
/* wire (canonical LE bytes x||y; (0,0) = ZERO, weierstrass.ts:716) <-> point */
static void PFX(pt_from_wire)(const PFX(ctx) * c, PFX(pt) * r, const uint8_t* w) {
  PFX(fe) x, y;
  memcpy(&x, w, NL * 8);
  memcpy(&y, w + NL * 8, NL * 8);
  if (PFX(is0)(&x) && PFX(is0)(&y)) {
    PFX(pt_zero)(c, r);
    return;
  }
  PFX(tomont)(c, &r->X, &x);
  PFX(tomont)(c, &r->Y, &y);
  r->Z = c->r1;
}
static int PFX(pt_to_wire)(const PFX(ctx) * c, uint8_t* w, const PFX(pt) * p) {
  PFX(fe) x, y;
  int inf = PFX(pt_to_affine)(c, &x, &y, p);
  PFX(frommont)(c, &x, &x);
  PFX(frommont)(c, &y, &y);
  memcpy(w, &x, NL * 8);
  memcpy(w + NL * 8, &y, NL * 8);
  return inf;
}

/* curve.ts:863-905 pippenger: unsigned windows, digit 0 not skipped, serial running sum. */
static void PFX(pippenger)(const PFX(ctx) * c, PFX(pt) * out, const PFX(pt) * pts, const uint8_t* scalars, size_t n,
                           int fn_bits) {
  PFX(pt) zero;
  PFX(pt_zero)(c, &zero);
  if (n == 0) {
    *out = zero;
    return;
  }
  int wbits = 0;
  for (size_t t = n; t; t >>= 1) wbits++;          /* bitLen(plength), curve.ts:879 */
  int ws = 1;
  if (wbits > 12) ws = wbits - 3;
  else if (wbits > 4) ws = wbits - 2;
  else if (wbits > 0) ws = 2;
  size_t nb = (size_t)1 << ws;
  uint64_t mask = nb - 1;
  PFX(pt)* buckets = (PFX(pt)*)malloc(nb * sizeof(PFX(pt)));
  int lastBits = ((fn_bits - 1) / ws) * ws;
  PFX(pt) sum = zero;
  for (int i = lastBits; i >= 0; i -= ws) {
    for (size_t j = 0; j < nb; j++) buckets[j] = zero;
    for (size_t j = 0; j < n; j++) {
      const uint8_t* s = scalars + 32 * j;          /* (scalar >> i) & MASK */
      uint64_t d = 0;
      for (int b = 0; b < ws; b++) {
        int bit = i + b;
        if (bit < 256) d |= (uint64_t)((s[bit >> 3] >> (bit & 7)) & 1) << b;
      }
      d &= mask;
      PFX(pt_add)(c, &buckets[d], &buckets[d], &pts[j]);
    }
    PFX(pt) resI = zero, sumI = zero;
    for (size_t j = nb - 1; j > 0; j--) {
      PFX(pt_add)(c, &sumI, &sumI, &buckets[j]);
      PFX(pt_add)(c, &resI, &resI, &sumI);
    }
    PFX(pt_add)(c, &sum, &sum, &resI);
    if (i != 0)
      for (int j = 0; j < ws; j++) PFX(pt_double)(c, &sum, &sum);
  }
  free(buckets);
  *out = sum;
}

/* curve.ts:431-447 wnafDigits(n, 4) over a little-endian byte scalar; returns digit count */
static int PFX(wnaf4)(int8_t* d, const uint8_t* k, int nbytes) {
  uint32_t w[12] = {0};                              /* up to 352 bits + headroom */
  for (int i = 0; i < nbytes; i++) w[i >> 2] |= (uint32_t)k[i] << (8 * (i & 3));
  int len = 0;
  for (;;) {
    int nz = 0;
    for (int i = 0; i < 12; i++) nz |= w[i] != 0;
    if (!nz) break;
    int dig = 0;
    if (w[0] & 1) {
      dig = (int)(w[0] & 15);
      if (dig >= 8) dig -= 16;
      /* n -= dig */
      int64_t carry = -(int64_t)dig;
      for (int i = 0; i < 12 && carry; i++) {
        int64_t t = (int64_t)w[i] + carry;
        w[i] = (uint32_t)t;
        carry = t >> 32;
      }
    }
    d[len++] = (int8_t)dig;
    for (int i = 0; i < 11; i++) w[i] = (w[i] >> 1) | (w[i + 1] << 31);
    w[11] >>= 1;
  }
  return len;
}

/* curve.ts:820-836 mulAddUnsafe over `np` (point, scalar) pairs: oddMultiples(p,4) tables
 * (:420-425), width-4 wNAF digits, shared-doubling walk (:479-498). scalars: nbytes LE each. */
static void PFX(mul_add_unsafe)(const PFX(ctx) * c, PFX(pt) * out, const PFX(pt) * pts, const uint8_t* scalars,
                                int nbytes, int np) {
  PFX(pt) tables[4][4];
  int8_t digits[4][400];
  int lens[4], mx = 0;
  for (int i = 0; i < np; i++) {
    PFX(pt) dbl;
    PFX(pt_double)(c, &dbl, &pts[i]);
    tables[i][0] = pts[i];
    for (int j = 1; j < 4; j++) PFX(pt_add)(c, &tables[i][j], &tables[i][j - 1], &dbl);
    lens[i] = PFX(wnaf4)(digits[i], scalars + (size_t)i * nbytes, nbytes);
    if (lens[i] > mx) mx = lens[i];
  }
  PFX(pt) acc;
  PFX(pt_zero)(c, &acc);
  for (int bit = mx - 1; bit >= 0; bit--) {
    if (bit != mx - 1) PFX(pt_double)(c, &acc, &acc);
    for (int i = 0; i < np; i++) {
      int w = bit < lens[i] ? digits[i][bit] : 0;
      if (w) {
        PFX(pt) item = tables[i][((w < 0 ? -w : w) - 1) >> 1];
        if (w < 0) PFX(pt_neg)(c, &item, &item);
        PFX(pt_add)(c, &acc, &acc, &item);
      }
    }
  }
  *out = acc;
}
