// hiprz_sort.hip — ray reordering between passes: the shade kernel's per-pixel keys -> a permutation the next trace kernel
// (and, from its own keys, the shadow kernel) follows.
#include <hipcub/hipcub.hpp>

#include "hiprz_ctx.hpp"

namespace hiprz {

int sort_workspace(hiprz_ctx* c, size_t n) {
    size_t bytes = 0;
    RZ_HIP(c, hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, c->sort_keys.ptr, c->sort_keys_out.ptr, c->sort_iota.ptr,
                                                 c->sort_perm.ptr, int(n), 0, 24, c->stream));
    RZ_HIP(c, c->sort_temp.resize(bytes));
    c->sort_temp_bytes = bytes;
    return HIPRZ_OK;
}

// radix sort of the keys the shade kernel just wrote -> permutation the next trace kernel follows
void launch_sort(hiprz_ctx* c) {
    if (!sort_enabled(c) || c->n_local_tiles == 0 || c->sorted_this_pass) return;
    c->sorted_this_pass = true;
    size_t bytes = c->sort_temp_bytes;
    (void)hipcub::DeviceRadixSort::SortPairs(c->sort_temp.ptr, bytes, c->sort_keys.ptr, c->sort_keys_out.ptr, c->sort_iota.ptr,
                                             c->sort_perm.ptr, int(c->n_local_tiles * 256u), 24 - effective_sort_bits(c), 24, c->stream);
}

// the same for the keys of the pass's shadow rays -> the order rz_shadow_kernel follows
void launch_shadow_sort(hiprz_ctx* c) {
    if (c->n_local_tiles == 0) return;
    size_t bytes = c->sort_temp_bytes;
    (void)hipcub::DeviceRadixSort::SortPairs(c->sort_temp.ptr, bytes, c->shadow_keys.ptr, c->sort_keys_out.ptr, c->sort_iota.ptr,
                                             c->shadow_perm.ptr, int(c->n_local_tiles * 256u), 24 - effective_sort_bits(c), 24, c->stream);
}

}  // namespace hiprz
