// SHA-256 of a byte string assembled from up to three pieces (prefix || middle || message), one lane per item:
// the prehash of ecdsa.verify (weierstrass.ts:1465-1470 validateMsgAndHash -> hash(message)) and the BIP-340 tagged
// hash sha256(tag || tag || r || pk || m) (src/secp256k1.ts:129-137).  The reference takes SHA-256 from
// @noble/hashes 2.3.0 (not vendored): this is FIPS 180-4 section 6.2 restated; parity is pinned by hashlib in the
// tests and by the reference's own ECDSA / BIP-340 vectors, which only verify if the hash is right.
#pragma once
#include "fp.hpp"

namespace ncg {

struct Sha256Consts {
  static constexpr uint32_t K[64] = {
      0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u, 0xd807aa98u, 0x12835b01u,
      0x243185beu, 0x550c7dc3u, 0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u, 0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu,
      0x2de92c6fu, 0x4a7484aau, 0x5cb0a9dcu, 0x76f988dau, 0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u,
      0x06ca6351u, 0x14292967u, 0x27b70a85u, 0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u, 0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u,
      0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u, 0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u, 0x19a4c116u, 0x1e376c08u,
      0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu, 0x682e6ff3u, 0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u,
      0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};
};

NCG_DI uint32_t sha_rotr32(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

// one compression of `w` (16 big-endian words, clobbered) into the state
NCG_DI void sha256_block(uint32_t (&h)[8], uint32_t (&w)[16]) {
  uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
  for (int t = 0; t < 64; t++) {
    uint32_t wt;
    if (t < 16) {
      wt = w[t];
    } else {
      const uint32_t w15 = w[(t - 15) & 15], w2 = w[(t - 2) & 15];
      const uint32_t s0 = sha_rotr32(w15, 7) ^ sha_rotr32(w15, 18) ^ (w15 >> 3);
      const uint32_t s1 = sha_rotr32(w2, 17) ^ sha_rotr32(w2, 19) ^ (w2 >> 10);
      wt = w[t & 15] + s0 + w[(t - 7) & 15] + s1;
      w[t & 15] = wt;
    }
    const uint32_t S1 = sha_rotr32(e, 6) ^ sha_rotr32(e, 11) ^ sha_rotr32(e, 25);
    const uint32_t ch = (e & f) ^ (~e & g);
    const uint32_t t1 = hh + S1 + ch + Sha256Consts::K[t] + wt;
    const uint32_t S0 = sha_rotr32(a, 2) ^ sha_rotr32(a, 13) ^ sha_rotr32(a, 22);
    const uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
    const uint32_t t2 = S0 + mj;
    hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
  }
  h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}

// digest (8 big-endian words) of pre[0..pre_len) || mid[0..mid_len) || msg[0..len)
NCG_DI void sha256_3(uint32_t (&h)[8], const uint8_t* __restrict__ pre, uint32_t pre_len, const uint8_t* __restrict__ mid,
                     uint32_t mid_len, const uint8_t* __restrict__ msg, uint64_t len) {
  h[0] = 0x6a09e667u; h[1] = 0xbb67ae85u; h[2] = 0x3c6ef372u; h[3] = 0xa54ff53au;
  h[4] = 0x510e527fu; h[5] = 0x9b05688cu; h[6] = 0x1f83d9abu; h[7] = 0x5be0cd19u;
  const uint64_t head = (uint64_t)pre_len + mid_len;
  const uint64_t total = head + len;
  const uint64_t nblocks = (total + 1 + 8 + 63) / 64;  // 0x80, 64-bit length, padding
  for (uint64_t blk = 0; blk < nblocks; blk++) {
    uint32_t w[16];
#pragma unroll 1
    for (int i = 0; i < 16; i++) {
      uint32_t v = 0;
      for (int j = 0; j < 4; j++) {
        const uint64_t pos = blk * 64 + (uint64_t)i * 4 + j;
        uint32_t byte;
        if (pos < pre_len) byte = pre[pos];
        else if (pos < head) byte = mid[pos - pre_len];
        else if (pos < total) byte = msg[pos - head];
        else if (pos == total) byte = 0x80u;
        else byte = 0;
        v = (v << 8) | byte;
      }
      w[i] = v;
    }
    if (blk == nblocks - 1) {  // bit length, big-endian 64-bit
      w[14] = (uint32_t)(total >> 29);
      w[15] = (uint32_t)(total << 3);
    }
    sha256_block(h, w);
  }
}

}  // namespace ncg
