/* Exhaustive check of the identities brisk_device_detect.h relies on where the reference divides a float by a constant in
 * double precision and rounds the quotient to float:
 *     (float)((double)v / c) == v / (float)c      for EVERY finite float v,  c = 6, 18, 3072
 * (the double quotient of a 24-bit value by these constants is never within a double rounding of a float rounding
 * boundary unless it is that boundary exactly, so rounding twice equals rounding once).
 * gcc -O2 -o /tmp/vfd tools/verify_float_division.c && /tmp/vfd [stride]      (stride 1 = all 2^32 bit patterns) */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int main(int argc, char** argv) {
  const uint64_t stride = argc > 1 ? strtoull(argv[1], 0, 10) : 1;
  const double cd[3] = {6.0, 18.0, 3072.0};
  const float cf[3] = {6.0f, 18.0f, 3072.0f};
  uint64_t bad[3] = {0, 0, 0}, n = 0;
  for (uint64_t b = 0; b < (1ull << 32); b += stride) {
    const uint32_t u = (uint32_t)b;
    float v;
    memcpy(&v, &u, 4);
    if (!isfinite(v)) continue;
    ++n;
    for (int k = 0; k < 3; ++k) {
      volatile double qd = (double)v / cd[k];
      const float a = (float)qd;
      volatile float q = v / cf[k];
      const float bq = q;
      if (memcmp(&a, &bq, 4) != 0) {
        if (bad[k]++ < 5) printf("c = %g: v = %a: %a != %a\n", cd[k], v, a, bq);
      }
    }
  }
  printf("%llu finite floats checked (stride %llu): mismatches %llu / %llu / %llu for c = 6 / 18 / 3072\n", (unsigned long long)n,
         (unsigned long long)stride, (unsigned long long)bad[0], (unsigned long long)bad[1], (unsigned long long)bad[2]);
  return (bad[0] | bad[1] | bad[2]) ? 1 : 0;
}
