// hiprz_sort.hip — ray reordering between passes, hand-written (no library kernels on the path).
//
// The shade kernel leaves a 24-bit key per pixel (cell of the next ray's origin + its direction, hiprz_device.hpp: ray_sort_key); a
// least-significant-digit radix sort over the key bits that matter (16 or 24, 8 bits per pass) turns the keys into the permutation the
// next trace kernel follows.  The deferred shadow rays get a permutation from their own keys the same way.
// (Gathering the rays themselves into sorted order — a contiguous 32-byte-per-ray stream for the trace kernel — was measured and
// dropped: the scattered 40-byte reads cost 108 us per pass on config C as a kernel of their own, 127 us fused into the last scatter,
// and saved the walk 21 us; inside the walk they hide behind its own latency.)
//
// One digit pass = count (per-tile histogram in LDS, digit totals) -> offsets (one workgroup per digit) -> scatter.  The scatter ranks keys STABLY
// inside a tile without per-key atomics: a round takes 256 consecutive keys; the lanes of a wave that hold the same digit find each
// other with 8 ballots (one per digit bit), the wave's per-digit counts meet in LDS, and thread d keeps digit d's running offset.
//
// Why not the library sort (hipcub / rocPRIM onesweep, round 1): inside a captured graph its replays faulted ("write access to a
// read-only page") as soon as another HIP user of the process — torch allocating a tensor between two batches — had been active,
// and launched eagerly it cost config C 15 % (ten launches per pass with their dispatch gaps).  These kernels take plain device
// pointers and sizes and are captured like the pass kernels.
// A single counting pass over 2^16..2^20 buckets with global atomics (histogram in the shade kernel, slots claimed by
// wave-aggregated atomicAdd) was measured too: C 4.2 -> 6.2 ms per step, E 51 -> 76 ms — a wave meets dozens of distinct buckets and
// a returning atomic per bucket serialises on its latency.
#include "hiprz_ctx.hpp"

namespace hiprz {
namespace {

constexpr uint32_t kTile = 4096u;       // keys per workgroup (256 threads x 16 rounds)
constexpr uint32_t kTotalCopies = 16u;  // copies of the per-digit totals the count kernel's atomics are spread over

__global__ void __launch_bounds__(256) rz_radix_count_kernel(const uint32_t* keys, uint32_t n, uint32_t shift, uint32_t* counts, uint32_t n_tiles, uint32_t* digit_total) {
    __shared__ uint32_t hist[256];
    hist[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t base = blockIdx.x * kTile + threadIdx.x;
    uint32_t key[16];
#pragma unroll
    for (uint32_t r = 0; r < 16u; ++r) key[r] = base + r * 256u < n ? keys[base + r * 256u] : 0u;
#pragma unroll
    for (uint32_t r = 0; r < 16u; ++r)
        if (base + r * 256u < n) atomicAdd(&hist[(key[r] >> shift) & 255u], 1u);
    __syncthreads();
    counts[threadIdx.x * n_tiles + blockIdx.x] = hist[threadIdx.x];  // digit-major: the counts of one digit over the tiles are contiguous
    // the digit totals in kTotalCopies interleaved copies (tile t adds to copy t % kTotalCopies): two thousand tiles adding to the same 256
    // words serialise in L2 — the offsets kernel sums the copies
    if (hist[threadIdx.x]) atomicAdd(&digit_total[(blockIdx.x % kTotalCopies) * 256u + threadIdx.x], hist[threadIdx.x]);
}

RZ_DEV uint32_t wave_inclusive_scan(uint32_t v) {
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t up = __shfl_up(v, off);
        if (int(threadIdx.x & 63u) >= off) v += up;
    }
    return v;
}
// counts -> offsets, one launch: workgroup d turns digit d's per-tile counts into exclusive offsets and adds the keys of all lower
// digits (the digit totals the count kernel accumulated; every workgroup sums the ones below its own digit itself).
__global__ void __launch_bounds__(256) rz_radix_offsets_kernel(uint32_t* counts, uint32_t n_tiles, const uint32_t* digit_total) {
    __shared__ uint32_t wave_total[4];
    const uint32_t digit = blockIdx.x, tid = threadIdx.x;
    uint32_t below = 0u;
    if (tid < digit)
        for (uint32_t k = 0; k < kTotalCopies; ++k) below += digit_total[k * 256u + tid];
    for (int off = 32; off > 0; off >>= 1) below += __shfl_down(below, off);
    if ((tid & 63u) == 0u) wave_total[tid >> 6] = below;
    __syncthreads();
    uint32_t carry = wave_total[0] + wave_total[1] + wave_total[2] + wave_total[3];
    __syncthreads();
    uint32_t* row = counts + size_t(digit) * n_tiles;
    for (uint32_t base = 0u; base < n_tiles; base += 256u) {
        const uint32_t i = base + tid, v = i < n_tiles ? row[i] : 0u;
        const uint32_t incl = wave_inclusive_scan(v);
        if ((tid & 63u) == 63u) wave_total[tid >> 6] = incl;
        __syncthreads();
        uint32_t before = carry;
        for (uint32_t w = 0; w < (tid >> 6); ++w) before += wave_total[w];
        if (i < n_tiles) row[i] = before + incl - v;
        carry += wave_total[0] + wave_total[1] + wave_total[2] + wave_total[3];
        __syncthreads();
    }
}

// FIRST: values are the pixel indices themselves (no value array to read).
template <bool FIRST>
__global__ void __launch_bounds__(256) rz_radix_scatter_kernel(const uint32_t* keys_in, const uint32_t* vals_in, uint32_t* keys_out, uint32_t* vals_out, uint32_t n,
                                                                uint32_t shift, const uint32_t* offsets, uint32_t n_tiles, uint32_t* digit_total) {
    __shared__ uint32_t run[256];           // where the next key of digit d goes
    __shared__ uint32_t wcount[2][4][256];  // keys of digit d held by wave w in this round; double-buffered by round (two barriers per round)
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    // the tile's 16 rounds of keys (and values) are fetched up front: sixteen loads in flight instead of one per barrier-bounded round
    uint32_t key[16], val[16];
#pragma unroll
    for (uint32_t r = 0; r < 16u; ++r) {
        const uint32_t i = blockIdx.x * kTile + r * 256u + tid;
        key[r] = i < n ? keys_in[i] : 0u;
        if constexpr (FIRST) val[r] = i;
        else val[r] = i < n ? vals_in[i] : 0u;
    }
    run[tid] = offsets[tid * n_tiles + blockIdx.x];
    if (blockIdx.x < kTotalCopies) digit_total[blockIdx.x * 256u + tid] = 0u;  // consumed by rz_radix_offsets_kernel; the next pass's count kernel adds to zeros
    for (uint32_t k = tid; k < 2048u; k += 256u) (&wcount[0][0][0])[k] = 0u;
    __syncthreads();
#pragma unroll
    for (uint32_t r = 0; r < 16u; ++r) {
        const uint32_t buf = r & 1u;
        const bool valid = blockIdx.x * kTile + r * 256u + tid < n;
        const uint32_t digit = (key[r] >> shift) & 255u;
        // the lanes of this wave with the same digit: one ballot per digit bit
        unsigned long long same = __ballot(valid);
#pragma unroll
        for (uint32_t b = 0; b < 8u; ++b) {
            const unsigned long long bit = __ballot((digit >> b) & 1u);
            same &= ((digit >> b) & 1u) ? bit : ~bit;
        }
        const uint32_t rank = uint32_t(__popcll(same & ((1ull << lane) - 1ull)));
        if (valid && rank == 0u) wcount[buf][wave][digit] = uint32_t(__popcll(same));
        __syncthreads();
        if (valid) {
            uint32_t pos = run[digit] + rank;  // stable: earlier rounds, then lower waves, then lower lanes
            for (uint32_t w = 0; w < wave; ++w) pos += wcount[buf][w][digit];
            if (keys_out) keys_out[pos] = key[r];
            if (vals_out) vals_out[pos] = val[r];
        }
        __syncthreads();
        // thread d owns digit d: advance its offset and clear this round's counters (the next round writes the other buffer)
        run[tid] += wcount[buf][0][tid] + wcount[buf][1][tid] + wcount[buf][2][tid] + wcount[buf][3][tid];
        wcount[buf][0][tid] = wcount[buf][1][tid] = wcount[buf][2][tid] = wcount[buf][3][tid] = 0u;
    }
}
__global__ void __launch_bounds__(256) rz_sort_identity_kernel(uint32_t n, uint32_t* perm) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) perm[i] = i;
}

// keys (destroyed) -> perm: least significant digit first, `passes` digits of 8 bits from bit `first_shift` up; `sorted_keys` (may be
// null) receives the keys in sorted order
void radix_sort_n(hipStream_t stream, uint32_t* keys, uint32_t n, uint32_t first_shift, int passes, uint32_t* perm, uint32_t* sorted_keys,
                  hiprz_frame_state::SortTemp& t) {
    const uint32_t n_tiles = (n + kTile - 1u) / kTile;
    uint32_t* key_buf[2] = {keys, t.keys_out.ptr};
    uint32_t* val_buf[2] = {t.vals_a.ptr, t.vals_b.ptr};
    for (int p = 0; p < passes; ++p) {
        const uint32_t shift = first_shift + 8u * uint32_t(p);
        const bool last = p + 1 == passes;
        const uint32_t* kin = key_buf[p & 1];
        uint32_t* kout = last ? sorted_keys : key_buf[(p + 1) & 1];
        const uint32_t* vin = val_buf[p & 1];
        uint32_t* vout = last ? perm : val_buf[(p + 1) & 1];
        RZ_LAUNCH(rz_radix_count_kernel, dim3(n_tiles), dim3(256), 0, stream, kin, n, shift, t.counts.ptr, n_tiles, t.digit_total.ptr);
        RZ_LAUNCH(rz_radix_offsets_kernel, dim3(256), dim3(256), 0, stream, t.counts.ptr, n_tiles, t.digit_total.ptr);
        if (p == 0) RZ_LAUNCH((rz_radix_scatter_kernel<true>), dim3(n_tiles), dim3(256), 0, stream, kin, vin, kout, vout, n, shift, t.counts.ptr, n_tiles, t.digit_total.ptr);
        else RZ_LAUNCH((rz_radix_scatter_kernel<false>), dim3(n_tiles), dim3(256), 0, stream, kin, vin, kout, vout, n, shift, t.counts.ptr, n_tiles, t.digit_total.ptr);
    }
}
void radix_sort(hiprz_ctx* c, uint32_t* keys, uint32_t* perm, hiprz_frame_state::SortTemp& t, hipStream_t stream) {
    const int passes = (effective_sort_bits(c) + 7) / 8;
    radix_sort_n(stream, keys, c->n_local_tiles * 256u, uint32_t(24 - 8 * passes), passes, perm, nullptr, t);
}

}  // namespace

int sort_temp_resize(hiprz_ctx* c, hiprz_frame_state::SortTemp& t, size_t n) {
    const size_t n_tiles = (n + kTile - 1u) / kTile, n_counts = 256u * n_tiles;
    RZ_HIP(c, t.keys_out.resize(n));
    RZ_HIP(c, t.vals_a.resize(n));  // value buffers of the middle passes
    RZ_HIP(c, t.vals_b.resize(n));
    RZ_HIP(c, t.counts.resize(n_counts));
    RZ_HIP(c, t.digit_total.resize(256u * kTotalCopies));  // keys per digit of the pass being sorted (zeroed again by its scatter kernel)
    RZ_HIP(c, hipMemsetAsync(t.digit_total.ptr, 0, 256u * kTotalCopies * sizeof(uint32_t), c->stream));
    return HIPRZ_OK;
}
int sort_workspace(hiprz_ctx* c, size_t n) {
    for (auto& t : c->sort_temp) {
        const int rc = sort_temp_resize(c, t, n);
        if (rc != HIPRZ_OK) return rc;
    }
    c->perm_valid = false;
    return HIPRZ_OK;
}
// any n keys of up to 32 bits on a stream of the caller's (the device-side tree build sorts Morton codes with it)
void sort_u32(hipStream_t stream, uint32_t* keys, uint32_t n, int key_bits, uint32_t* perm, uint32_t* sorted_keys, hiprz_frame_state::SortTemp& t) {
    radix_sort_n(stream, keys, n, 0u, (key_bits + 7) / 8, perm, sorted_keys, t);
}

// The keys the shade kernel just wrote -> the order of the next pass's rays.  `beside`: on the auxiliary stream, after everything the
// main stream has been given so far — the deferred shadow-ray kernel that follows on the main stream does not need this order, and
// the sort's small, bandwidth-light kernels fit beside its VALU-bound walk; join_sort() makes the main stream wait for it.
void launch_sort(hiprz_ctx* c, bool beside) {
    if (!sort_enabled(c) || c->n_local_tiles == 0 || c->sorted_this_pass) return;
    c->sorted_this_pass = true;
    if (beside && c->aux_stream) {
        (void)hipEventRecord(c->aux_fork, c->stream);
        (void)hipStreamWaitEvent(c->aux_stream, c->aux_fork, 0);
        radix_sort(c, c->sort_keys.ptr, c->sort_perm.ptr, c->sort_temp[0], c->aux_stream);
        (void)hipEventRecord(c->aux_join, c->aux_stream);
        c->sort_beside = true;
    } else {
        radix_sort(c, c->sort_keys.ptr, c->sort_perm.ptr, c->sort_temp[0], c->stream);
    }
    c->perm_valid = true;
}
void join_sort(hiprz_ctx* c) {
    if (!c->sort_beside) return;
    (void)hipStreamWaitEvent(c->stream, c->aux_join, 0);
    c->sort_beside = false;
}

// rays that have not been through a sort (reordering was switched on between two batches): the identity order
void launch_sort_identity(hiprz_ctx* c) {
    if (c->n_local_tiles == 0) return;
    RZ_LAUNCH(rz_sort_identity_kernel, dim3(c->n_local_tiles), dim3(256), 0, c->stream, c->n_local_tiles * 256u, c->sort_perm.ptr);
    c->perm_valid = true;
}

// the same for the keys of the pass's shadow rays -> the order the shadow kernel follows
void launch_shadow_sort(hiprz_ctx* c) {
    if (c->n_local_tiles == 0) return;
    radix_sort(c, c->shadow_keys.ptr, c->shadow_perm.ptr, c->sort_temp[1], c->stream);
}

}  // namespace hiprz
