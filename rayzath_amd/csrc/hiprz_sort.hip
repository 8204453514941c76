// hiprz_sort.hip — ray reordering between passes, hand-written (no library kernels on the path).
//
// The shade kernel leaves a 24-bit key per pixel (the cell of the next ray's origin interleaved with the cell where it leaves the world box,
// hiprz_device.hpp: ray_sort_key); a
// least-significant-digit radix sort over the key bits that matter (16 or 24, 8 bits per pass) turns the keys into the permutation the
// next trace kernel follows.  The deferred shadow rays get a permutation from their own keys the same way.
// (Gathering the rays themselves into sorted order — a contiguous 32-byte-per-ray stream for the trace kernel — was measured and
// dropped: the scattered 40-byte reads cost 108 us per pass on config C as a kernel of their own, 127 us fused into the last scatter,
// and saved the walk 21 us; inside the walk they hide behind its own latency.)
//
// One digit pass = count (per-tile histogram) -> offsets (one workgroup per digit scans its row of tile counts) -> scatter.  The scatter
// ranks keys STABLY inside a tile without per-key atomics: the four waves rank their own quarters of the tile — the lanes of a wave that
// hold the same digit find each other with 8 ballots (one per digit bit), the wave's running per-digit counts live in LDS — then the
// tile's keys meet in LDS in digit order and consecutive threads write consecutive keys of one digit to consecutive addresses.
// (Round 4; until then every key went from its register straight to its place: 4-byte writes to 256 places per tile.  8.3 M keys of
// 24 bits: 405 -> 194 us, `tools/sort_bench.py`; config E's step 42.6 -> 41.3 ms.)
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
// Per-tile digit counts (the digit totals come out of the offsets kernel's scan: no global atomics).  Every wave counts its quarter of
// the tile into a histogram of its own; up to four digits that the wave's 64 keys share are counted by one lane each — sixty-four
// rays of one cell would serialise on one LDS word — and whatever is left goes through LDS atomics.
__global__ void __launch_bounds__(256) rz_radix_count_kernel(const uint32_t* keys, uint32_t n, uint32_t shift, uint32_t* counts, uint32_t n_tiles) {
    __shared__ uint32_t wcnt[4][256];
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    wcnt[0][tid] = wcnt[1][tid] = wcnt[2][tid] = wcnt[3][tid] = 0u;
    const uint32_t first = blockIdx.x * kTile + wave * 1024u + lane;
    uint32_t key[16];
#pragma unroll
    for (uint32_t r = 0; r < 16u; ++r) key[r] = first + r * 64u < n ? keys[first + r * 64u] : 0u;
    __syncthreads();
#pragma unroll
    for (uint32_t r = 0; r < 16u; ++r) {
        const bool valid = first + r * 64u < n;
        const uint32_t digit = (key[r] >> shift) & 255u;
        unsigned long long left = __ballot(valid);
#pragma unroll
        for (int peel = 0; peel < 4; ++peel) {
            if (left == 0ull) break;
            const uint32_t d = uint32_t(__builtin_amdgcn_readlane(int(digit), int(__builtin_ctzll(left))));
            const unsigned long long group = __ballot(digit == d) & left;
            if (lane == 0u) wcnt[wave][d] += uint32_t(__popcll(group));
            left &= ~group;
        }
        if ((left >> lane) & 1ull) atomicAdd(&wcnt[wave][digit], 1u);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    counts[tid * n_tiles + blockIdx.x] = wcnt[0][tid] + wcnt[1][tid] + wcnt[2][tid] + wcnt[3][tid];  // digit-major: the counts of one digit over the tiles are contiguous
}

RZ_DEV uint32_t wave_inclusive_scan(uint32_t v) {
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t up = __shfl_up(v, off);
        if (int(threadIdx.x & 63u) >= off) v += up;
    }
    return v;
}
// The scatter with its writes in runs: the waves rank their own quarters of the tile (no workgroup barrier inside the ranking — a
// wave's running per-digit counts are its own), the tile's keys meet in LDS in digit order, and consecutive threads then write
// consecutive keys of one digit to consecutive addresses.  Stable: a digit's keys keep the order they came in (earlier wave quarter,
// earlier round, lower lane).  FIRST: values are the key indices themselves (no value array to read).
template <bool FIRST>
__global__ void __launch_bounds__(256) rz_radix_scatter_kernel(const uint32_t* keys_in, const uint32_t* vals_in, uint32_t* keys_out, uint32_t* vals_out, uint32_t n,
                                                                     uint32_t shift, const uint32_t* offsets, uint32_t n_tiles, const uint32_t* row_total) {
    __shared__ uint32_t wcnt[4][256];   // ranking: keys of digit d wave w has met so far; then: where wave w's keys of digit d start in the staged tile
    __shared__ uint32_t gbase[256];     // global position of the tile's first key of digit d, minus that key's staged slot
    __shared__ uint32_t wtot[4], wbelow[4];
    __shared__ uint32_t stage_key[kTile], stage_val[kTile];
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    const uint32_t tile_base = blockIdx.x * kTile, first = tile_base + wave * 1024u + lane;
    uint32_t key[16], val[16];
#pragma unroll
    for (uint32_t r = 0; r < 16u; ++r) {
        const uint32_t i = first + r * 64u;
        key[r] = i < n ? keys_in[i] : 0u;
        if constexpr (FIRST) val[r] = i;
        else val[r] = i < n ? vals_in[i] : 0u;
    }
    const uint32_t tile_offset = offsets[tid * n_tiles + blockIdx.x], digit_keys = row_total[tid];
    wcnt[0][tid] = wcnt[1][tid] = wcnt[2][tid] = wcnt[3][tid] = 0u;
    __syncthreads();
    uint32_t rank[16];
#pragma unroll
    for (uint32_t r = 0; r < 16u; ++r) {
        const bool valid = first + r * 64u < n;
        const uint32_t digit = (key[r] >> shift) & 255u;
        unsigned long long same = __ballot(valid);
#pragma unroll
        for (uint32_t b = 0; b < 8u; ++b) {
            const unsigned long long bit = __ballot((digit >> b) & 1u);
            same &= ((digit >> b) & 1u) ? bit : ~bit;
        }
        const uint32_t before = uint32_t(__popcll(same & ((1ull << lane) - 1ull)));
        const uint32_t met = wcnt[wave][digit];
        rank[r] = met + before;
        __builtin_amdgcn_wave_barrier();
        if (valid && before == 0u) wcnt[wave][digit] = met + uint32_t(__popcll(same));
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    {   // thread d: digit d's keys in this tile, its exclusive start among the digits, the waves' starts inside its run
        const uint32_t c0 = wcnt[0][tid], c1 = wcnt[1][tid], c2 = wcnt[2][tid], c3 = wcnt[3][tid], total = c0 + c1 + c2 + c3;
        const uint32_t incl = wave_inclusive_scan(total), below_incl = wave_inclusive_scan(digit_keys);
        if (lane == 63u) wtot[wave] = incl, wbelow[wave] = below_incl;
        __syncthreads();
        uint32_t start = incl - total, below = below_incl - digit_keys;  // keys of lower digits: in this tile, in the whole array
        for (uint32_t w = 0; w < wave; ++w) start += wtot[w], below += wbelow[w];
        wcnt[0][tid] = start, wcnt[1][tid] = start + c0, wcnt[2][tid] = start + c0 + c1, wcnt[3][tid] = start + c0 + c1 + c2;
        gbase[tid] = below + tile_offset - start;
    }
    __syncthreads();
#pragma unroll
    for (uint32_t r = 0; r < 16u; ++r) {
        if (first + r * 64u < n) {
            const uint32_t slot = wcnt[wave][(key[r] >> shift) & 255u] + rank[r];
            stage_key[slot] = key[r], stage_val[slot] = val[r];
        }
    }
    __syncthreads();
    const uint32_t here = n - tile_base < kTile ? n - tile_base : kTile;
#pragma unroll
    for (uint32_t j = 0; j < 16u; ++j) {
        const uint32_t slot = j * 256u + tid;
        if (slot < here) {
            const uint32_t k = stage_key[slot], pos = gbase[(k >> shift) & 255u] + slot;
            if (keys_out) keys_out[pos] = k;
            vals_out[pos] = stage_val[slot];
        }
    }
}

// counts -> offsets inside one digit's row, every entry of the row in flight at once (thread t scans 8 consecutive tiles), and the
// row's total: the scatter adds the keys of all lower digits itself (a 256-entry scan per tile)
__global__ void __launch_bounds__(256) rz_radix_offsets_kernel(uint32_t* counts, uint32_t n_tiles, uint32_t* row_total) {
    __shared__ uint32_t wave_total[4];
    const uint32_t digit = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint32_t carry = 0u;
    uint32_t* row = counts + size_t(digit) * n_tiles;
    for (uint32_t base = 0u; base < n_tiles; base += 2048u) {
        const uint32_t at = base + tid * 8u;
        uint32_t v[8], sum = 0u;
#pragma unroll
        for (uint32_t k = 0; k < 8u; ++k) v[k] = at + k < n_tiles ? row[at + k] : 0u;
#pragma unroll
        for (uint32_t k = 0; k < 8u; ++k) {
            const uint32_t c = v[k];
            v[k] = sum, sum += c;
        }
        const uint32_t incl = wave_inclusive_scan(sum);
        if (lane == 63u) wave_total[wave] = incl;
        __syncthreads();
        uint32_t before = carry + incl - sum;
        for (uint32_t w = 0; w < wave; ++w) before += wave_total[w];
#pragma unroll
        for (uint32_t k = 0; k < 8u; ++k)
            if (at + k < n_tiles) row[at + k] = before + v[k];
        carry += wave_total[0] + wave_total[1] + wave_total[2] + wave_total[3];
        __syncthreads();
    }
    if (tid == 0u) row_total[digit] = carry;
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
        RZ_LAUNCH(rz_radix_count_kernel, dim3(n_tiles), dim3(256), 0, stream, kin, n, shift, t.counts.ptr, n_tiles);
        RZ_LAUNCH(rz_radix_offsets_kernel, dim3(256), dim3(256), 0, stream, t.counts.ptr, n_tiles, t.row_total.ptr);
        if (p == 0) RZ_LAUNCH((rz_radix_scatter_kernel<true>), dim3(n_tiles), dim3(256), 0, stream, kin, vin, kout, vout, n, shift, t.counts.ptr, n_tiles, t.row_total.ptr);
        else RZ_LAUNCH((rz_radix_scatter_kernel<false>), dim3(n_tiles), dim3(256), 0, stream, kin, vin, kout, vout, n, shift, t.counts.ptr, n_tiles, t.row_total.ptr);
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
    RZ_HIP(c, t.row_total.resize(256u));  // keys per digit of the pass being sorted (rz_radix_offsets_kernel -> rz_radix_scatter_kernel)
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

// hiprz.h: the sort checked directly — a stable permutation in key order — and timed on the device
int hiprz_selftest_sort(hiprz_ctx* c, const uint32_t* keys_host, uint32_t n, int key_bits, uint32_t repeats, uint64_t* errors, double* sort_us) {
    using namespace hiprz;
    if (!c || !errors || !sort_us || (n && !keys_host) || key_bits < 1 || key_bits > 32 || n > (1u << 30)) return HIPRZ_ERR_INVALID;
    *errors = 0, *sort_us = 0.0;
    if (n == 0u) return HIPRZ_OK;
    (void)hipSetDevice(c->device);
    struct Work {
        DeviceArray<uint32_t> keys, perm, sorted;
        hiprz_frame_state::SortTemp temp;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        ~Work() {
            keys.release(), perm.release(), sorted.release();
            temp.keys_out.release(), temp.vals_a.release(), temp.vals_b.release(), temp.counts.release(), temp.row_total.release();
            if (e0) (void)hipEventDestroy(e0);
            if (e1) (void)hipEventDestroy(e1);
        }
    } w;
    RZ_HIP(c, w.keys.resize(n));
    RZ_HIP(c, w.perm.resize(n));
    RZ_HIP(c, w.sorted.resize(n));
    const int rc = sort_temp_resize(c, w.temp, n);
    if (rc != HIPRZ_OK) return rc;
    RZ_HIP(c, hipEventCreate(&w.e0));
    RZ_HIP(c, hipEventCreate(&w.e1));
    double best = 0.0;
    for (uint32_t r = 0; r < (repeats ? repeats : 1u); ++r) {
        RZ_HIP(c, hipMemcpyAsync(w.keys.ptr, keys_host, sizeof(uint32_t) * n, hipMemcpyHostToDevice, c->stream));
        RZ_HIP(c, hipEventRecord(w.e0, c->stream));
        sort_u32(c->stream, w.keys.ptr, n, key_bits, w.perm.ptr, w.sorted.ptr, w.temp);
        RZ_HIP(c, hipEventRecord(w.e1, c->stream));
        RZ_HIP(c, hipStreamSynchronize(c->stream));
        float ms = 0.0f;
        RZ_HIP(c, hipEventElapsedTime(&ms, w.e0, w.e1));
        if (r == 0u || double(ms) < best) best = double(ms);
    }
    *sort_us = best * 1000.0;
    std::vector<uint32_t> perm(n), sorted(n);
    RZ_HIP(c, hipMemcpy(perm.data(), w.perm.ptr, sizeof(uint32_t) * n, hipMemcpyDeviceToHost));
    RZ_HIP(c, hipMemcpy(sorted.data(), w.sorted.ptr, sizeof(uint32_t) * n, hipMemcpyDeviceToHost));
    // what the digits the sort looks at leave of a key: whole bytes from bit 0 up
    const int passes = (key_bits + 7) / 8;
    const uint32_t seen = passes >= 4 ? 0xFFFFFFFFu : ((1u << (8 * passes)) - 1u);
    std::vector<uint8_t> met(n, 0);
    uint64_t bad = 0;
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t at = perm[i];
        if (at >= n || met[at]) {
            ++bad;  // not a permutation
            continue;
        }
        met[at] = 1;
        if (sorted[i] != keys_host[at]) ++bad;  // the sorted keys are not the keys in the permutation's order
        if (i) {
            const uint32_t before = perm[i - 1] < n ? keys_host[perm[i - 1]] & seen : 0u, here = keys_host[at] & seen;
            if (before > here) ++bad;                               // not in key order
            else if (before == here && perm[i - 1] >= at) ++bad;    // equal keys out of their original order: not stable
        }
    }
    *errors = bad;
    return HIPRZ_OK;
}

