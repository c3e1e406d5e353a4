// devcache.h -- process-wide cache of the device blocks of resident problems (defined in solver.hip).
//
// Incremental Bundler calls run_sfm hundreds of times on problems of 14 .. 400 cameras (src/BundleFast.cpp:263-438); every call
// builds a problem of ~60 device buffers and tears it down again.  hipMalloc is cheap on this runtime, hipFree is not: each one
// synchronises the device and gives the pages back (40-60 us apiece: 0.7 ms of a 4 ms call at 14 cameras, 2.5 of 16 at 50,
// profiles/r03_small_problem_latency.txt).  Blocks therefore go back to a free list keyed by (device, size class) -- eight classes
// per octave, at most 12.5 % of slack -- and the next problem of a similar size takes them from there.  Nothing is zeroed on reuse
// (hipMalloc does not promise zeros either).  The list is bounded (BSFM_DEVCACHE_MB, default 6144; 0 disables it), is emptied when
// an allocation fails, and bsfm_device_cache_trim() empties it on request.
#pragma once
#include <hip/hip_runtime_api.h>
#include <cstddef>

namespace bsfm {
// size in bytes; *p is device memory of at least that size
hipError_t dev_alloc(void** p, size_t bytes);
// synced = the caller has synchronised the device since the block was last used (bsfm_problem_destroy does, once, for all of its
// blocks); otherwise this call synchronises first, like hipFree.  Pointers the cache did not hand out are passed on to hipFree.
void dev_free(void* p, bool synced = false);
}
