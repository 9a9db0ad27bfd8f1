// exhaustive check of the reciprocal form of x / w used by the t-statistics (events.hip, TStatAt): for w = 2..64, every float x
// with 2^-100 <= |x| <= 2^100:  fma(fma(-q, w, x), rw, q) with q = x * rw, rw = RN(1 / w)  ==  x / w bit for bit.
// gcc -O2 -ffp-contract=off -mfma -fopenmp -include stdlib.h tools/micro/div_by_window_check.c -lm; ./a.out 2 64  (1 min on 8 cores: 0 mismatches)
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
int main(int argc, char **argv) {
  int w0 = argc > 1 ? atoi(argv[1]) : 2, w1 = argc > 2 ? atoi(argv[2]) : 16;
  unsigned long long total_bad = 0;
  for (int w = w0; w <= w1; ++w) {
    const float wf = (float)w, rw = 1.0f / wf;
    uint64_t bad = 0;
    float lo = 0x1p-100f, hi = 0x1p100f;
    uint32_t lb, hb; memcpy(&lb, &lo, 4); memcpy(&hb, &hi, 4);
#pragma omp parallel for reduction(+:bad) schedule(static)
    for (int64_t u = lb; u <= (int64_t)hb; ++u) {
      for (int s = 0; s < 2; ++s) {
        uint32_t ub = (uint32_t)u | ((uint32_t)s << 31);
        float x; memcpy(&x, &ub, 4);
        const float q = x * rw;
        const float r = fmaf(-q, wf, x);
        const float q2 = fmaf(r, rw, q);
        const float want = x / wf;
        uint32_t a, b; memcpy(&a, &q2, 4); memcpy(&b, &want, 4);
        bad += a != b;
      }
    }
    printf("w %d: mismatches %llu\n", w, (unsigned long long)bad);
    fflush(stdout);
    total_bad += bad;
  }
  return total_bad ? 1 : 0;
}
