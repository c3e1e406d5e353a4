/* oracle/match_oracle.c -- TEST INFRASTRUCTURE ONLY (CPU oracle "port" of the matcher).
 *
 * Restates MatchKeys (src/keys2a.cpp:347-372) with the kd-tree replaced by the exhaustive search it
 * approximates: for every key of set 1 the two nearest keys of set 2 in squared L2 (int32, max
 * 128*255^2 = 8 323 200 as in lib/ann_1.1_char/include/ANN/ANN.h:161-162), kept iff
 * (double) d0 < ratio*ratio*(double) d1  (keys2a.cpp:362).  With the ANN visit cap disabled and eps = 0
 * the reference returns exactly this set, which tests/test_matcher.py checks against
 * oracle/_ref/libkeymatchref.so.  Ties: the lowest index wins the nearest slot; a tie between the two
 * nearest can never pass the strict test, so the tie order is unobservable.
 */
#include <limits.h>

int oracle_match_keys(int n1, const unsigned char *k1, int n2, const unsigned char *k2, double ratio,
                      int *out_pairs, int max_out)
{
    int i, j, q, cnt = 0;
    if (n2 < 2) return -1;
    for (i = 0; i < n1; i++) {
        const unsigned char *a = k1 + 128 * (long) i;
        int d0 = INT_MAX, d1 = INT_MAX, i0 = -1;
        for (j = 0; j < n2; j++) {
            const unsigned char *b = k2 + 128 * (long) j;
            int d = 0;
            for (q = 0; q < 128; q++) { int t = (int) a[q] - (int) b[q]; d += t * t; }
            if (d < d0) { d1 = d0; d0 = d; i0 = j; }
            else if (d < d1) d1 = d;
        }
        if (((double) d0) < ratio * ratio * ((double) d1)) {
            if (cnt < max_out) { out_pairs[2 * cnt] = i; out_pairs[2 * cnt + 1] = i0; }
            cnt++;
        }
    }
    return cnt;
}
