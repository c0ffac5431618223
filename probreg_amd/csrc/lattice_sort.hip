// Stable LSD radix sort of (vertex id, incidence) pairs for the order-preserving splat of the permutohedral lattice
// (filterreg.hip, lat_segments).  The sort itself is rocPRIM's device-wide radix sort - a plain library primitive, in a
// translation unit of its own so that its templates are instantiated once.
#include <rocprim/device/device_radix_sort.hpp>

#include "prg_common.h"

namespace prg {

// Sorts n (key, value) pairs by the low `bits` bits of the key; equal keys keep their input order.  tmp == nullptr: only
// *tmp_bytes is set (workspace query).
int sort_pairs_u32(void* tmp, size_t* tmp_bytes, const unsigned* keys_in, unsigned* keys_out, const int* vals_in,
                   int* vals_out, unsigned n, unsigned bits, hipStream_t stream) {
    size_t bytes = *tmp_bytes;
    PRG_HIP(rocprim::radix_sort_pairs(tmp, bytes, keys_in, keys_out, vals_in, vals_out, n, 0u, bits, stream));
    *tmp_bytes = bytes;
    return PRG_OK;
}

}  // namespace prg
