// SHA-512 of R || A || M and reduction mod L, one lane per signature: the challenge of
// eddsa.verify, k = modN_LE(SHA-512(dom2 || R || A || M)) with an empty domain for pure ed25519
// (src/abstract/edwards.ts:984 `hashDomainToScalar`, :900-906, modN_LE :866-868).  The reference takes
// SHA-512 from @noble/hashes 2.3.0 (not vendored in the repository): this is FIPS 180-4 section 6.4
// restated; parity is pinned by the reference's own vectors (test/vectors/ed25519/vectors.txt and
// zip215.json verify only if the hash is right) and by hashlib in the tests.
#pragma once
#include "fp.hpp"
#include "scalar.hpp"

namespace ncg {

struct Sha512Consts {
  static constexpr uint64_t K[80] = {
      0x428a2f98d728ae22ull, 0x7137449123ef65cdull, 0xb5c0fbcfec4d3b2full, 0xe9b5dba58189dbbcull, 0x3956c25bf348b538ull,
      0x59f111f1b605d019ull, 0x923f82a4af194f9bull, 0xab1c5ed5da6d8118ull, 0xd807aa98a3030242ull, 0x12835b0145706fbeull,
      0x243185be4ee4b28cull, 0x550c7dc3d5ffb4e2ull, 0x72be5d74f27b896full, 0x80deb1fe3b1696b1ull, 0x9bdc06a725c71235ull,
      0xc19bf174cf692694ull, 0xe49b69c19ef14ad2ull, 0xefbe4786384f25e3ull, 0x0fc19dc68b8cd5b5ull, 0x240ca1cc77ac9c65ull,
      0x2de92c6f592b0275ull, 0x4a7484aa6ea6e483ull, 0x5cb0a9dcbd41fbd4ull, 0x76f988da831153b5ull, 0x983e5152ee66dfabull,
      0xa831c66d2db43210ull, 0xb00327c898fb213full, 0xbf597fc7beef0ee4ull, 0xc6e00bf33da88fc2ull, 0xd5a79147930aa725ull,
      0x06ca6351e003826full, 0x142929670a0e6e70ull, 0x27b70a8546d22ffcull, 0x2e1b21385c26c926ull, 0x4d2c6dfc5ac42aedull,
      0x53380d139d95b3dfull, 0x650a73548baf63deull, 0x766a0abb3c77b2a8ull, 0x81c2c92e47edaee6ull, 0x92722c851482353bull,
      0xa2bfe8a14cf10364ull, 0xa81a664bbc423001ull, 0xc24b8b70d0f89791ull, 0xc76c51a30654be30ull, 0xd192e819d6ef5218ull,
      0xd69906245565a910ull, 0xf40e35855771202aull, 0x106aa07032bbd1b8ull, 0x19a4c116b8d2d0c8ull, 0x1e376c085141ab53ull,
      0x2748774cdf8eeb99ull, 0x34b0bcb5e19b48a8ull, 0x391c0cb3c5c95a63ull, 0x4ed8aa4ae3418acbull, 0x5b9cca4f7763e373ull,
      0x682e6ff3d6b2b8a3ull, 0x748f82ee5defb2fcull, 0x78a5636f43172f60ull, 0x84c87814a1f0ab72ull, 0x8cc702081a6439ecull,
      0x90befffa23631e28ull, 0xa4506cebde82bde9ull, 0xbef9a3f7b2c67915ull, 0xc67178f2e372532bull, 0xca273eceea26619cull,
      0xd186b8c721c0c207ull, 0xeada7dd6cde0eb1eull, 0xf57d4f7fee6ed178ull, 0x06f067aa72176fbaull, 0x0a637dc5a2c898a6ull,
      0x113f9804bef90daeull, 0x1b710b35131c471bull, 0x28db77f523047d84ull, 0x32caab7b40c72493ull, 0x3c9ebe0a15c9bebcull,
      0x431d67c49c100d4cull, 0x4cc5d4becb3e42b6ull, 0x597f299cfc657e2aull, 0x5fcb6fab3ad6faecull, 0x6c44198c4a475817ull};
  // L = 2^252 + DELTA (the ed25519 group order), DELTA as 4 LE limbs
  static constexpr uint32_t DELTA[4] = {0x5cf5d3edu, 0x5812631au, 0xa2f79cd6u, 0x14def9deu};
};

NCG_DI uint64_t sha_rotr(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }

// one compression of `w` (16 big-endian words, clobbered) into the state
NCG_DI void sha512_block(uint64_t (&h)[8], uint64_t (&w)[16]) {
  uint64_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
  for (int t = 0; t < 80; t++) {
    uint64_t wt;
    if (t < 16) {
      wt = w[t];
    } else {
      const uint64_t w15 = w[(t - 15) & 15], w2 = w[(t - 2) & 15];
      const uint64_t s0 = sha_rotr(w15, 1) ^ sha_rotr(w15, 8) ^ (w15 >> 7);
      const uint64_t s1 = sha_rotr(w2, 19) ^ sha_rotr(w2, 61) ^ (w2 >> 6);
      wt = w[t & 15] + s0 + w[(t - 7) & 15] + s1;
      w[t & 15] = wt;
    }
    const uint64_t S1 = sha_rotr(e, 14) ^ sha_rotr(e, 18) ^ sha_rotr(e, 41);
    const uint64_t ch = (e & f) ^ (~e & g);
    const uint64_t t1 = hh + S1 + ch + Sha512Consts::K[t] + wt;
    const uint64_t S0 = sha_rotr(a, 28) ^ sha_rotr(a, 34) ^ sha_rotr(a, 39);
    const uint64_t mj = (a & b) ^ (a & c) ^ (b & c);
    const uint64_t t2 = S0 + mj;
    hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
  }
  h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}

// digest (64 bytes as 8 big-endian words -> the byte string) of r32 || a32 || msg[0..len)
NCG_DI void sha512_ram(uint64_t (&h)[8], const uint8_t* __restrict__ r32, const uint8_t* __restrict__ a32,
                       const uint8_t* __restrict__ msg, uint64_t len) {
  h[0] = 0x6a09e667f3bcc908ull; h[1] = 0xbb67ae8584caa73bull; h[2] = 0x3c6ef372fe94f82bull; h[3] = 0xa54ff53a5f1d36f1ull;
  h[4] = 0x510e527fade682d1ull; h[5] = 0x9b05688c2b3e6c1full; h[6] = 0x1f83d9abfb41bd6bull; h[7] = 0x5be0cd19137e2179ull;
  const uint64_t total = 64 + len;                       // bytes hashed
  const uint64_t nblocks = (total + 1 + 16 + 127) / 128;  // 0x80, 128-bit length, padding
  for (uint64_t blk = 0; blk < nblocks; blk++) {
    uint64_t w[16];
#pragma unroll 1
    for (int i = 0; i < 16; i++) {
      uint64_t v = 0;
      for (int j = 0; j < 8; j++) {
        const uint64_t pos = blk * 128 + (uint64_t)i * 8 + j;
        uint32_t byte;
        if (pos < 32) byte = r32[pos];
        else if (pos < 64) byte = a32[pos - 32];
        else if (pos < total) byte = msg[pos - 64];
        else if (pos == total) byte = 0x80u;
        else byte = 0;
        v = (v << 8) | byte;
      }
      w[i] = v;
    }
    if (blk == nblocks - 1) {  // bit length, big-endian 128-bit (high half is zero for any real message)
      w[14] = total >> 61;
      w[15] = total << 3;
    }
    sha512_block(h, w);
  }
}

// k = LE(digest) mod L as 8 LE limbs.  L = 2^252 + DELTA: fold 2^252 = -DELTA three times (the third
// fold multiplies at most 6 bits), then bring the signed remainder into [0, L).
NCG_DI void mod_l_512(uint32_t (&k)[8], const uint32_t (&x)[16]) {  // k = x mod L, x < 2^512 as 16 LE limbs
  uint32_t delta[4];
#pragma unroll
  for (int i = 0; i < 4; i++) delta[i] = Sha512Consts::DELTA[i];
  // split at bit 252: lo0 (8 limbs, top limb 28 bits), hi0 (260 bits -> 9 limbs)
  uint32_t lo0[9], hi0[9];
#pragma unroll
  for (int i = 0; i < 8; i++) lo0[i] = x[i];
  lo0[7] &= 0x0fffffffu;
  lo0[8] = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) hi0[i] = (x[7 + i] >> 28) | (i + 8 < 16 ? (x[8 + i] << 4) : 0u);
  // t = hi0 * DELTA  (13 limbs, < 2^385)
  uint32_t t[13];
  mp_mul<9, 4>(t, hi0, delta);
  uint32_t tlo[9], thi[5];
#pragma unroll
  for (int i = 0; i < 8; i++) tlo[i] = t[i];
  tlo[7] &= 0x0fffffffu;
  tlo[8] = 0;
#pragma unroll
  for (int i = 0; i < 5; i++) thi[i] = (t[7 + i] >> 28) | (i + 8 < 13 ? (t[8 + i] << 4) : 0u);
  // u = thi * DELTA (9 limbs, < 2^258)
  uint32_t u[9];
  mp_mul<5, 4>(u, thi, delta);
  uint32_t ulo[9];
#pragma unroll
  for (int i = 0; i < 8; i++) ulo[i] = u[i];
  ulo[7] &= 0x0fffffffu;
  ulo[8] = 0;
  const uint32_t uhi[1] = {(u[7] >> 28) | (u[8] << 4)};  // < 2^7
  uint32_t v5[5];
  mp_mul<1, 4>(v5, uhi, delta);  // < 2^132
  uint32_t v[9];
#pragma unroll
  for (int i = 0; i < 9; i++) v[i] = i < 5 ? v5[i] : 0u;
  // x = lo0 - tlo + ulo - v  (mod L), each term below 2^252: result in (-2^253, 2^253), 9-limb two's complement
  uint32_t r[9], s[9];
  mp_sub<9>(r, lo0, tlo);
  mp_add<9>(s, r, ulo);
  mp_sub<9>(r, s, v);
  uint32_t lmod[9];
#pragma unroll
  for (int i = 0; i < 9; i++) lmod[i] = i < 4 ? delta[i] : 0u;
  lmod[7] = 0x10000000u;  // L
  // add L while negative (at most twice), then subtract L while >= L (at most twice)
#pragma unroll
  for (int pass = 0; pass < 2; pass++) {
    const bool neg = (r[8] >> 31) != 0;
    uint32_t a2[9];
    mp_add<9>(a2, r, lmod);
#pragma unroll
    for (int i = 0; i < 9; i++) r[i] = neg ? a2[i] : r[i];
  }
#pragma unroll
  for (int pass = 0; pass < 2; pass++) {
    uint32_t d2[9];
    const uint32_t bw = mp_sub<9>(d2, r, lmod);
    const bool ge = bw == 0;
#pragma unroll
    for (int i = 0; i < 9; i++) r[i] = ge ? d2[i] : r[i];
  }
#pragma unroll
  for (int i = 0; i < 8; i++) k[i] = r[i];
}
NCG_DI void sha512_digest_mod_l(uint32_t (&k)[8], const uint64_t (&h)[8]) {
  // digest bytes are the big-endian words h[0..7]; as a little-endian integer, limb j (32-bit) is
  // bytes 4j..4j+3: byte-swapped halves of h[j/2]
  uint32_t x[16];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const uint32_t hi = (uint32_t)(h[i] >> 32), lo = (uint32_t)h[i];
    x[2 * i] = __builtin_bswap32(hi);
    x[2 * i + 1] = __builtin_bswap32(lo);
  }
  mod_l_512(k, x);
}

}  // namespace ncg
