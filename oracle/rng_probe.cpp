// Prints libstdc++'s std::minstd_rand / std::uniform_real_distribution<double>
// stream for a seed: the very objects the reference's sampler holds
// (inst/include/glmmrmcml/mhmcmc.h:27-28, seeded at :55, drawn at :85).
// Test infrastructure: tests/test_oracle_rng.py checks oracle/mcml_oracle.c's
// restatement against this, bit for bit.
#include <cstdio>
#include <cstdlib>
#include <random>
int main(int argc, char** argv) {
  unsigned seed = argc > 1 ? (unsigned)strtoul(argv[1], nullptr, 10) : 12345u;
  int n = argc > 2 ? atoi(argv[2]) : 8;
  std::minstd_rand gen(seed);
  std::uniform_real_distribution<double> dist(0.0, 1.0);
  for (int i = 0; i < n; i++) printf("%a\n", dist(gen));
  std::minstd_rand g1(1);
  unsigned x = 0;
  for (int i = 0; i < 10000; i++) x = g1();
  printf("kat10000 %u\n", x);
  return 0;
}
