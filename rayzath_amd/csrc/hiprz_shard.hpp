// hiprz_shard.hpp — which shard owns which 32x8-pixel tile, and where the shard keeps it (hiprz_set_shard).
//
// The tiles of the frame are numbered row by row, t = 0 .. tiles_x * tiles_y - 1; shard s of `world` owns the numbers t with
// t % world == s and keeps them in order (local tile lt = t / world), so the shards' tile counts differ by at most one.  What is NOT plain
// row-major is where number t lies in its row: row r is rotated by shift(r) columns — number r * tiles_x + k is the tile in column
// (k + shift(r)) % tiles_x — with shift(r) chosen so that column c of row r goes to shard (c + offset(r)) % world, where offset() runs
// through a permutation of 0 .. world - 1 every `world` rows.
//
// The rotation is what spreads a thin feature of the image over the shards.  Unrotated, a column of tiles falls on the shards
// (r * tiles_x + c) % world: on 2 of 8 shards at 1920 pixels (60 tiles per row, 60 % 8 = 4) and on ONE of 8 at 1280, 2560 or 3840 (tiles per
// row a multiple of 8) — a lamp post, a door frame or the edge of a wall would be rendered by one GPU.  With the offsets a column is dealt to
// all shards in turn, and (for 2, 4, 8 shards: found by exhaustive search over the permutations) no line of slope dy/dx with
// |dx|, |dy| <= 4 tiles puts more than twice its fair share on one shard.  (The first shift(r) < world columns of a row wrap around the
// row's end and follow the unrotated rule.)  With world == 1 the rotation is 0: plain row-major tile order.
//
// A padded variant — every shard tiles_y * ceil(tiles_x / world) local tiles, those beyond the right edge empty — was measured first and
// dropped: an eighth of a 1080p frame then is 1080 workgroups instead of 1013, more than the 1024 that are resident at once (4 per CU), and
// the resident kernels' step went from 0.30 to 0.41 ms (config B) although the empty workgroups left at once (profiles/r03/ab_shard_map.txt).
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define RZ_SHARD_FN __host__ __device__ inline
#else
#define RZ_SHARD_FN inline
#endif

namespace hiprz {

RZ_SHARD_FN uint32_t shard_row_offset(uint32_t row, uint32_t world) {
    if (world == 8u) return (0x62457310u >> (4u * (row & 7u))) & 7u;  // 0 1 3 7 5 4 2 6
    if (world == 4u) return (0x2310u >> (4u * (row & 3u))) & 3u;      // 0 1 3 2
    return row % world;
}
// columns by which tile row `row` is rotated: (row * tiles_x + c - shift) % world == (c + offset(row)) % world
RZ_SHARD_FN uint32_t shard_row_shift(uint32_t row, uint32_t tiles_x, uint32_t world) {
    return ((row % world) * (tiles_x % world) + world - shard_row_offset(row, world)) % world % tiles_x;
}
RZ_SHARD_FN uint32_t shard_local_tiles(uint32_t tiles_x, uint32_t tiles_y, uint32_t rank, uint32_t world) {
    const uint32_t n_tiles = tiles_x * tiles_y;
    return rank < n_tiles ? (n_tiles - rank + world - 1u) / world : 0u;
}
// local tile lt of shard (rank, world) -> its place in the grid
RZ_SHARD_FN void shard_tile(uint32_t lt, uint32_t tiles_x, uint32_t rank, uint32_t world, uint32_t& tx, uint32_t& ty) {
    const uint32_t t = lt * world + rank;
    ty = t / tiles_x;
    tx = t - ty * tiles_x + shard_row_shift(ty, tiles_x, world);
    if (tx >= tiles_x) tx -= tiles_x;
}
// tile (tx, ty) -> the shard that owns it and its local index there
RZ_SHARD_FN void shard_of_tile(uint32_t tx, uint32_t ty, uint32_t tiles_x, uint32_t world, uint32_t& rank, uint32_t& lt) {
    const uint32_t shift = shard_row_shift(ty, tiles_x, world);
    const uint32_t t = ty * tiles_x + (tx >= shift ? tx - shift : tx + tiles_x - shift);
    rank = t % world, lt = t / world;
}

}  // namespace hiprz
