// oracle/ref_keymatch.cpp -- TEST INFRASTRUCTURE ONLY.
// C wrapper around the REFERENCE'S OWN matcher (src/keys2a.cpp:326-372 + lib/ann_1.1_char), compiled
// from /root/reference by oracle/Makefile into oracle/_ref/libkeymatchref.so.
//   max_pts_visit = 200 : KeyMatchFull as shipped (src/KeyMatchFull.cpp:126-127) -- timing baseline.
//   max_pts_visit = 0   : visit cap disabled (lib/ann_1.1_char/src/kd_pr_search.cpp:115-116) with eps=0
//                         => exact 2-NN, the bit-exact equality target for the HIP brute-force kernel.
#include <vector>
#include <ctime>
#include "keys2a.h"

extern "C" int ref_match_keys(int n1, unsigned char *k1, int n2, unsigned char *k2, double ratio,
                              int max_pts_visit, int *out_pairs, int max_out, double *secs)
{
    clock_t t0 = clock();
    ANNkd_tree *tree = CreateSearchTree(n2, k2);
    std::vector<KeypointMatch> mt = MatchKeys(n1, k1, tree, ratio, max_pts_visit);
    clock_t t1 = clock();
    if (secs) *secs = (double) (t1 - t0) / CLOCKS_PER_SEC;
    int cnt = (int) mt.size();
    for (int i = 0; i < cnt && i < max_out; i++) {
        out_pairs[2 * i] = mt[i].m_idx1;
        out_pairs[2 * i + 1] = mt[i].m_idx2;
    }
    // the reference never frees the ANN point array either (src/KeyMatchFull.cpp:149)
    { ANNpointArray pa = tree->thePoints(); annDeallocPts(pa); }
    delete tree;
    return cnt;
}
