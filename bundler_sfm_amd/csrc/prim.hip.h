// prim.hip.h -- the device-wide primitives of the one-time index construction (index_build.hip, tracks.hip), called on rocPRIM
// DIRECTLY (round 2 went through the hipcub headers, a CUB-compatibility facade over the same library).  Two-phase convention of
// rocPRIM: a call with tmp == nullptr only reports the temporary-storage size.
#pragma once
#include <hip/hip_runtime.h>
#include <cstring>
#include <iterator>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_run_length_encode.hpp>
#include <rocprim/device/device_select.hpp>

namespace bsfm { namespace prim {

// stable LSD radix sort of (key, value) pairs on key bits [begin_bit, end_bit)
template <typename K, typename V>
inline hipError_t sort_pairs(void* tmp, size_t& bytes, const K* keys_in, K* keys_out, const V* vals_in, V* vals_out, size_t n,
                             int begin_bit, int end_bit, hipStream_t st)
{
    return rocprim::radix_sort_pairs(tmp, bytes, keys_in, keys_out, vals_in, vals_out, n, (unsigned)begin_bit, (unsigned)end_bit, st);
}
template <typename K>
inline hipError_t sort_keys(void* tmp, size_t& bytes, const K* keys_in, K* keys_out, size_t n, int begin_bit, int end_bit, hipStream_t st)
{
    return rocprim::radix_sort_keys(tmp, bytes, keys_in, keys_out, n, (unsigned)begin_bit, (unsigned)end_bit, st);
}
// out[i] = in[0] + ... + in[i-1]
template <typename T>
inline hipError_t exclusive_sum(void* tmp, size_t& bytes, const T* in, T* out, size_t n, hipStream_t st)
{
    return rocprim::exclusive_scan(tmp, bytes, in, out, T(0), n, rocprim::plus<T>(), st);
}
// runs of equal keys: unique keys, run lengths, number of runs
template <typename K, typename C>
inline hipError_t run_length_encode(void* tmp, size_t& bytes, const K* in, K* unique_out, C* counts_out, int* nruns_out, size_t n, hipStream_t st)
{
    return rocprim::run_length_encode(tmp, bytes, in, (unsigned int)n, unique_out, counts_out, nruns_out, st);
}
// out = the items of `in` whose flag is non-zero, order kept; *count_out = how many
template <typename T, typename F>
inline hipError_t select_flagged(void* tmp, size_t& bytes, const T* in, const F* flags, T* out, int* count_out, size_t n, hipStream_t st)
{
    return rocprim::select(tmp, bytes, in, flags, out, count_out, n, st);
}

} }  // namespace bsfm::prim
