// hiprz_launch_batch.hip — the fused pass kernel (one kernel per pass; the resident pipeline's renderFirstPass) and the
// resident batch kernel (ONE launch takes every owned tile through all the cumulative passes of a render call).
#include "hiprz_ctx.hpp"
#include "hiprz_kernels.hpp"

namespace hiprz {
namespace {

template <bool FIRST, bool COUNT>
void launch_fused_t(hiprz_ctx* c, const DFrame& f) {
    const PassGeometry g = pass_geometry(c);
    const DConfig cfg = make_config(c);
    if (c->mode_flags & kIntegratorFlags) {  // CUDA-compat mode: its own fused kernel on the global scene
        RZ_LAUNCH((rz_compat_pass_kernel<FIRST, COUNT>), g.grid, g.block, 0, c->stream, c->dscene, c->dcamera, cfg, f);
        return;
    }
    // the fused kernel's shadow rays use the stack walk: its columns must exist in every mode
    const size_t lds = g.mode == 2 ? g.walk_lds + 4096u : g.walk_lds;  // mode 2: + the parked path state
    if (g.mode == 2) {
        if (g.lds_scene && c->flat_world) RZ_LAUNCH((rz_pass_kernel<FIRST, COUNT, 4, true>), g.grid, g.block, g.blob + lds, c->stream, c->dscene, c->dcamera, cfg, f);
        else if (g.lds_scene) RZ_LAUNCH((rz_pass_kernel<FIRST, COUNT, 2, true>), g.grid, g.block, g.blob + lds, c->stream, c->dscene, c->dcamera, cfg, f);
        else RZ_LAUNCH((rz_pass_kernel<FIRST, COUNT, 2, false>), g.grid, g.block, lds, c->stream, c->dscene, c->dcamera, cfg, f);
    } else {
        if (g.lds_scene) RZ_LAUNCH((rz_pass_kernel<FIRST, COUNT, 1, true>), g.grid, g.block, g.blob + lds, c->stream, c->dscene, c->dcamera, cfg, f);
        else RZ_LAUNCH((rz_pass_kernel<FIRST, COUNT, 1, false>), g.grid, g.block, lds, c->stream, c->dscene, c->dcamera, cfg, f);
    }
}

template <bool COUNT>
void launch_batch_t(hiprz_ctx* c, const DFrame& f, uint32_t n) {
    PassGeometry g = pass_geometry(c);
    const DConfig cfg = make_config(c);
    // counted renders report the work of the reference's visiting order unless asked otherwise (hiprz_set_walk_order): those keep the
    // workgroup kernel with its stack walk in that order
    const bool reference_counters = COUNT && c->walk_order != 2 && c->scene_tree == HIPRZ_TREE_REFERENCE;
    if (wave_resident(c) && reference_counters) g.mode = 1, g.walk_lds = g.stack_lds;
    if (wave_resident(c) && !reference_counters) {  // scenes that are not staged in LDS, without lights: single-wave workgroups walk cooperatively, pass after pass
        const dim3 wgrid(c->n_local_tiles * 4u), wblock(64);
        const bool one_leaf_world = c->dscene.n_instances != 0u && c->flat_world;  // (hiprz_launch_trace.hip: the plain one-step world level)
        if (c->n_textures == 0u && one_leaf_world) RZ_LAUNCH((rz_wave_batch_kernel<COUNT, RZ_SHADOW_PLAIN, 4, true>), wgrid, wblock, CoopLds::kBytes, c->stream, c->dscene, c->dcamera, cfg, f, n);
        else if (c->n_textures == 0u) RZ_LAUNCH((rz_wave_batch_kernel<COUNT, RZ_SHADOW_PLAIN, 4>), wgrid, wblock, CoopLds::kBytes, c->stream, c->dscene, c->dcamera, cfg, f, n);
        else if (one_leaf_world) RZ_LAUNCH((rz_wave_batch_kernel<COUNT, RZ_SHADOW_NONE, 4, true>), wgrid, wblock, CoopLds::kBytes, c->stream, c->dscene, c->dcamera, cfg, f, n);
        else RZ_LAUNCH((rz_wave_batch_kernel<COUNT, RZ_SHADOW_NONE, 4>), wgrid, wblock, CoopLds::kBytes, c->stream, c->dscene, c->dcamera, cfg, f, n);
        return;
    }
    const dim3 grid = g.grid, block = g.block;
    const size_t park = 8u * 1024u;
    const size_t lds = g.blob + g.walk_lds + park;
    const uint32_t park_offset = uint32_t(g.walk_lds);
    // scenes without lights run the instantiation whose next-event-estimation code is compiled out (RZ_SHADOW_NONE), scenes that
    // have no maps either the one without texture fetches and normal mapping (RZ_SHADOW_PLAIN)
    const bool dark = c->dscene.n_spot_lights + c->dscene.n_direct_lights == 0u && c->nolight_kernels;
    const bool plain = dark && c->n_textures == 0u;
    // 5 workgroups per CU must fit LDS, and the grid must be more than two full loads of the chip (256 CUs x 5)
    const bool five = lds * 5u <= 160u * 1024u && grid.x > 2u * 5u * 256u && c->batch_waves != 4;
#define RZ_BATCH(M, L)                                                                                                                     \
    do {                                                                                                                                   \
        if (plain && five) RZ_LAUNCH((rz_batch_kernel<COUNT, M, L, RZ_SHADOW_PLAIN, 5>), grid, block, lds, c->stream, c->dscene, c->dcamera, cfg, f, n, park_offset); \
        else if (plain) RZ_LAUNCH((rz_batch_kernel<COUNT, M, L, RZ_SHADOW_PLAIN>), grid, block, lds, c->stream, c->dscene, c->dcamera, cfg, f, n, park_offset); \
        else if (dark) RZ_LAUNCH((rz_batch_kernel<COUNT, M, L, RZ_SHADOW_NONE>), grid, block, lds, c->stream, c->dscene, c->dcamera, cfg, f, n, park_offset); \
        else RZ_LAUNCH((rz_batch_kernel<COUNT, M, L, 1>), grid, block, lds, c->stream, c->dscene, c->dcamera, cfg, f, n, park_offset);     \
    } while (0)
    if (g.mode == 2) {
        if (g.lds_scene && c->flat_world) RZ_BATCH(4, true);  // a one-leaf world: instance boxes tested up front
        else if (g.lds_scene) RZ_BATCH(2, true);
        else RZ_BATCH(2, false);
    } else {
        if (g.lds_scene) RZ_BATCH(1, true);
        else RZ_BATCH(1, false);
    }
#undef RZ_BATCH
}

}  // namespace

void launch_fused(hiprz_ctx* c, const DFrame& f, bool first, bool counted) {
    if (first) counted ? launch_fused_t<true, true>(c, f) : launch_fused_t<true, false>(c, f);
    else counted ? launch_fused_t<false, true>(c, f) : launch_fused_t<false, false>(c, f);
}

void launch_batch(hiprz_ctx* c, const DFrame& f, uint32_t n_passes, bool counted, hipEvent_t before, hipEvent_t after) {
    if (before) (void)hipEventRecord(before, c->stream);
    counted ? launch_batch_t<true>(c, f, n_passes) : launch_batch_t<false>(c, f, n_passes);
    if (after) (void)hipEventRecord(after, c->stream);
}

}  // namespace hiprz
