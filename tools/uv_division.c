/* Exhaustive check behind pixel_coord() of csrc/atmo_kernels.hip: for every viewport dimension n <= 65536 (the largest
 * atmo_render accepts) and every pixel index 0 <= i < n, ONE Markstein correction of (i + 0.5) * RN(1/n),
 *     q0 = a * r;  q = fma(fma(-q0, n, a), r, q0),      a = i + 0.5,  r = RN(1 / n),
 * equals the IEEE-754 binary32 quotient a / n.   gcc -O2 -o uv_division tools/uv_division.c -lm && ./uv_division
 * (2.1e9 quotients, under a minute; prints the number of mismatches, exit status 1 if any).  Also checks byte / 255 in the
 * two-instruction form of unorm8_exact(). */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int main(int argc, char **argv) {
    const int max_n = argc > 1 ? atoi(argv[1]) : 65536;  /* the tests run a prefix; the full range takes ~17 s */
    long bad = 0, total = 0;
    for (int n = 1; n <= max_n; ++n) {
        const float c = (float)n, r = 1.0f / c;
        for (int i = 0; i < n; ++i) {
            const float a = (float)i + 0.5f;
            const float q0 = a * r;
            if (fmaf(fmaf(-q0, c, a), r, q0) != a / c) ++bad;
        }
        total += n;
    }
    printf("pixel_coord: %ld mismatches in %ld quotients (n <= %d)\n", bad, total, max_n);
    long bad8 = 0;
    uint32_t hi = 0x3b808081u, lo = 0xaf7efeffu;
    float c_hi, c_lo;
    memcpy(&c_hi, &hi, 4);
    memcpy(&c_lo, &lo, 4);
    for (int b = 0; b < 256; ++b)
        if (fmaf((float)b, c_hi, (float)b * c_lo) != (float)b / 255.0f) ++bad8;
    printf("unorm8_exact: %ld mismatches in 256 bytes\n", bad8);
    return (bad || bad8) ? 1 : 0;
}
