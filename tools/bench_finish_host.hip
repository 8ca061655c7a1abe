// Host-side timing of the MSM finish (csrc/bls_host64.hpp: Horner over the grouped window sums + affine conversion) on random
// accumulators, and of its field product.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Inoble-curves_amd/csrc tools/bench_finish_host.hip -o tools/_build/bench_finish_host
#include <chrono>
#include <cstdio>
#include <cstring>
#include <cstdint>
#include <vector>
#include <random>
#include "bls_host64.hpp"
using namespace ncg;
int main() {
  std::mt19937_64 rng(1);
  const int nwin = 20, ng = 4, FW = 28, c = 13, g = 4;
  std::vector<uint32_t> fin((size_t)ng * nwin * 4 * FW);
  for (auto& x : fin) x = rng() & 0x1fffffff;
  uint32_t out[48]; uint8_t inf;
  for (int rep = 0; rep < 3; rep++) {
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 20; i++) h64::msm_finish<h64::Fp2>(fin.data(), c, nwin, g, ng, FW, 24, out, &inf);
    auto t1 = std::chrono::steady_clock::now();
    printf("G2 finish: %.1f us  (out %08x)\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / 20, out[3]);
  }
  {
    const int nwin1 = 16, ng1 = 5, FW1 = 14;
    for (int rep = 0; rep < 2; rep++) {
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 20; i++) h64::msm_finish<h64::Fp>(fin.data(), 16, nwin1, 4, ng1, FW1, 12, out, &inf);
    auto t1 = std::chrono::steady_clock::now();
    printf("G1 finish: %.1f us  (out %08x)\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / 20, out[3]);
    }
  }
  h64::Fp a = h64::from_fe29(fin.data()), b = h64::from_fe29(fin.data() + 14);
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < 1000000; i++) a = h64::mul(a, b);
  auto t1 = std::chrono::steady_clock::now();
  printf("dependent mul: %.1f ns (%llx)\n", std::chrono::duration<double, std::nano>(t1 - t0).count() / 1e6, (unsigned long long)a.v[0]);
  h64::Fp x[4] = {a, b, a, b};
  t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < 250000; i++) for (int j = 0; j < 4; j++) x[j] = h64::mul(x[j], b);
  t1 = std::chrono::steady_clock::now();
  printf("4 independent muls: %.1f ns each (%llx)\n", std::chrono::duration<double, std::nano>(t1 - t0).count() / 1e6, (unsigned long long)(x[0].v[0]^x[1].v[0]^x[2].v[0]^x[3].v[0]));
}
