/*
 * kmr_sort.hip -- the one device-wide sort of the library, in a translation unit of its own (rocPRIM's templates take a while to
 * compile): (u64 key, u32 value) pairs by key.  Used by the correction pass for k-mers seen more than 65 535 times
 * (finalize_superkmer_t: their sightings in stream order), a few pairs per build if any.
 */
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>

namespace kmr {
/* tmp == nullptr: only the size of the temporary storage is returned in *tmp_bytes */
int sort_pairs_u64_u32(void *tmp, size_t *tmp_bytes, const unsigned long long *keys_in, unsigned long long *keys_out, const unsigned int *vals_in, unsigned int *vals_out,
                       size_t n, hipStream_t stream) {
	return (int)rocprim::radix_sort_pairs(tmp, *tmp_bytes, keys_in, keys_out, vals_in, vals_out, n, 0, 64, stream);
}
}
