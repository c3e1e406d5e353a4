// match_l2.hip -- placeholder, filled in below (brute-force 128-D uchar L2 matcher).
#include "../../include/bsfm.h"
#include <cstdio>
extern "C" int bsfm_match_keys_l2(int, const unsigned char*, int, const unsigned char*, double, int*, int)
{ fprintf(stderr, "[bsfm] matcher not built yet\n"); return BSFM_ERROR; }
extern "C" int bsfm_key_match_full(int, const int*, const unsigned char* const*, double, int, const char*)
{ fprintf(stderr, "[bsfm] matcher not built yet\n"); return BSFM_ERROR; }
