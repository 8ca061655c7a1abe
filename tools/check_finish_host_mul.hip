#include <cstdio>
#include <random>
#include "bls_host64.hpp"
using namespace ncg;
int main() {
  std::mt19937_64 rng(7);
  printf("adx: %d\n", (int)h64::have_adx());
  const h64::Consts& k = h64::K();
  int bad = 0;
  for (int it = 0; it < 2000000; it++) {
    h64::Fp a, b;
    for (int i = 0; i < 6; i++) { a.v[i] = rng(); b.v[i] = rng(); }
    if (it % 7 == 0) for (int i = 0; i < 6; i++) a.v[i] = ~0ull;
    if (it % 11 == 0) for (int i = 0; i < 6; i++) b.v[i] = ~0ull;
    if (it % 13 == 0) for (int i = 0; i < 6; i++) a.v[i] = k.p[i] - (i == 0);
    if (it % 3) { a.v[5] &= (1ull << 61) - 1; b.v[5] &= (1ull << 60) - 1; }  // mostly < p
    // canonical operands (the domain): reduce below p
    if (it % 3) { while (h64::geq(a.v, k.p)) h64::sub_in_place(a.v, k.p); while (h64::geq(b.v, k.p)) h64::sub_in_place(b.v, k.p); }
    else continue;
#if NCG_H64_ADX
    h64::Fp x = h64::mul_c(a, b), y = h64::mul_adx(a, b);
#else
    h64::Fp x = h64::mul_c(a, b), y = x;
#endif
    if (!h64::eq(x, y)) { if (bad < 5) printf("MISMATCH it %d\n", it); bad++; }
  }
  printf("bad %d\n", bad);
  return bad != 0;
}
