// exhaustive check of StartCoef (chain.hip: the start score factor without the double division) against the division:
// all floats in [0, 8 r) for six radii.  gcc -O2 -ffp-contract=off -fopenmp tools/micro/start_coef_check.c -lm  (7 s on 8 cores: 0 mismatches)
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
#include <omp.h>
static inline float fast(float dist, double radius, double inv_radius, int *fell) {
  const double x = 0.2 * (double)dist;
  const double q = x * inv_radius, y = 1.0 - q;
  uint64_t b; memcpy(&b, &y, 8);
  const uint32_t low = (uint32_t)b & 0x1FFFFFFFu;
  const uint32_t off = low > 0x10000000u ? low - 0x10000000u : 0x10000000u - low;
  const double ay = fabs(y);
  const int safe = fabs(q) <= 4.0 * ay && off > 64u && ay >= 0x1p-100 && ay <= 0x1p100;
  if (safe) return (float)y;
  *fell = 1;
  return (float)(1 - x / radius);
}
int main(int argc, char **argv) {
  const float radii[] = {0.08f, 0.15f, 0.05f, 1.0f, 0.3333333f, 7.77f};
  const int n_radii = argc > 1 ? atoi(argv[1]) : 6;  // (tests/test_micro_checks.py runs the first one)
  uint64_t total_bad = 0;
  for (int r = 0; r < n_radii && r < 6; ++r) {
    const double R = (double)radii[r], Rinv = 1.0 / R;
    float lim = 8.0f * radii[r]; uint32_t lb; memcpy(&lb, &lim, 4);
    uint64_t bad = 0, fb = 0, maxoff = 0;
#pragma omp parallel for reduction(+:bad,fb)
    for (int64_t u = 0; u < (int64_t)lb; ++u) {
      uint32_t ub = (uint32_t)u; float d; memcpy(&d, &ub, 4);
      int fell = 0;
      const float a = fast(d, R, Rinv, &fell);
      const float w = (float)(1 - 0.2 * (double)d / R);
      uint32_t ab, wb; memcpy(&ab, &a, 4); memcpy(&wb, &w, 4);
      bad += ab != wb; fb += fell;
    }
    printf("radius %g: %u values, mismatches %llu, took the division %llu\n", radii[r], lb, (unsigned long long)bad, (unsigned long long)fb);
    total_bad += bad;
  }
  return total_bad ? 1 : 0;
}
