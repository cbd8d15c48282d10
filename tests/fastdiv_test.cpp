// tests/test_fastdiv.py: csrc/fpx_fastdiv.hpp (host side of the same source the kernels compile) against `/` and `%`
#include <cstdint>
#include <cstdio>
#include <random>

#include "../frankenpaxos_amd/csrc/fpx_fastdiv.hpp"

static long long checked = 0;

static bool check(int d, int s) {
  uint32_t m;
  int32_t sh;
  fpx::fast_div_magic(d, &m, &sh);
  ++checked;
  if (d < 2) return m == 0;
  const int q = fpx::fast_div(s, m, sh);
  if (q != s / d || s - q * d != s % d) {
    std::printf("MISMATCH d = %d s = %d: %d (magic %u shift %d), expected %d\n", d, s, q, m, sh, s / d);
    return false;
  }
  return true;
}

int main() {
  std::mt19937_64 rng(2026);
  const int top = 0x7fffffff;
  bool ok = check(1, 5) && check(0, 5);
  auto divisor = [&](int d) {
    // the edges of the dividend's range, the multiples of d next to them, and random dividends
    const int edge[] = {0, 1, d - 1, d, d + 1, 2 * d - 1, top, top - 1, top / d * d, top / d * d - 1, (top / d - 1) * d + d - 1};
    for (int s : edge)
      if (s >= 0) ok = check(d, s) && ok;
    for (int k = 0; k < 2000; ++k) ok = check(d, (int)(rng() & 0x7fffffffu)) && ok;
    for (int k = 0; k < 200; ++k) ok = check(d, (int)(rng() % 100000)) && ok;
  };
  for (int d = 2; d <= 4096; ++d) divisor(d);
  for (int k = 1; k < 31; ++k) {
    divisor(1 << k);
    if ((1 << k) > 2) divisor((1 << k) - 1);
    if (k < 30) divisor((1 << k) + 1);
  }
  divisor(top);
  for (int k = 0; k < 20000; ++k) divisor((int)(2 + rng() % (uint64_t)(top - 2)));
  // every slot of a window the size of the headline's, for the leader-group counts the configs use
  for (int d : {2, 3, 5, 7, 16, 255, 256, 1000})
    for (int s = 0; s < (1 << 22); ++s) ok = check(d, s) && ok;
  std::printf("%s: %lld divisions checked\n", ok ? "fastdiv ok" : "fastdiv FAILED", checked);
  return ok ? 0 : 1;
}
