// hiprz_launch_shade.hip — split pipeline, second half of a pass: everything of traceRay after the closest hit
// (cpu_engine_kernel.cpp:118-177) in rz_shade_kernel, and — for scenes with lights that are not staged in LDS — the pass's
// shadow rays (anyIntersection, :398-481) in a lean kernel of their own, walked in their own sorted order.
#include "hiprz_ctx.hpp"
#include "hiprz_kernels.hpp"

#ifndef RZ_PACKET_MINW
#define RZ_PACKET_MINW 4  // waves per SIMD the wave-level shadow walk's register budget is cut for (it takes 90 VGPRs: 5 waves)
#endif

namespace hiprz {
namespace {

// The two orders a shaded pass needs.  The next pass's ray order is needed by the shadow kernel only when it has no order of its own
// (HIPRZ_SHADOW_SORT=0); otherwise it is sorted on the auxiliary stream beside the shadow rays' sort and walk, and the main stream
// picks it up after them (join_sort).  (The shadow rays' sort first and alone, the ray sort beside the walk only: measured on E in
// round 4 with the runs sort, 41.8 against 41.4 ms per step, and again beside the wave-level shadow walk, 35.05 against 34.5 — two memory-bound sorts share the chip better than a sort and the walk.)
void sort_after_shading(hiprz_ctx* c, const DFrame& f) {
    launch_sort(c, f.shadow_key != nullptr);
    if (f.shadow_key) launch_shadow_sort(c);
}

// The deferred shadow rays in their own sorted order (slot set, light, origin cell) reach the kernel as BEAMS — 64 rays from one cell towards
// one light — and the wave walks the trees for all of them at once (rz_shadow_packet_kernel; round 4, config E: shadow kernel 1 449 ->
// about 1 160 us, step 37.5 -> 35.2 ms, identical frames).  Not for counted passes (the work counters are anchored on the per-lane walks)
// not where the shadow rays follow the next pass's ray order (HIPRZ_SHADOW_SORT=0: no beams), and not for small frames of many instances (wide
// beams: below).  HIPRZ_SHADOW_PACKET=0 / 1: never / always.
template <bool COUNT>
bool shadow_beams(const hiprz_ctx* c, const DFrame& f) {
    if (COUNT || c->shadow_packet == 0 || f.shadow_key == nullptr) return false;
    // how narrow the beams are goes with the rays per light and cell: the living room at 40 instances 1.09x (4K) / 1.06x (1080p) / 1.05x (960 x 540) /
    // 0.97x (480 x 270) of the cooperative walk's pass, at 300 instances 0.98x (1080p) / 0.88x (480 x 270) — tools/ab_shadow_walks.py
    return c->shadow_packet > 0 || size_t(c->n_local_tiles) * 256u >= size_t(8192) * c->dscene.n_instances;
}

template <bool FIRST, bool COUNT>
void launch_shade_t(hiprz_ctx* c, const DFrame& f) {
    const PassGeometry g = pass_geometry(c);
    const DConfig cfg = make_config(c);
    const dim3 grid = g.grid, block = g.block;
    const bool lights = c->dscene.n_spot_lights + c->dscene.n_direct_lights != 0u;
    if (c->mode_flags & kIntegratorFlags) {
        // CUDA-compat integrator: the same packaging as below — shading, then (scenes with lights) the pass's shadow rays in the lean
        // cooperative kernel in their own sorted order; with HIPRZ_COMPAT_SHADOW_COLOR its mask-collecting instantiation (round 4).
        if (lights && defer_shadows(c)) {
            RZ_LAUNCH((rz_shade_kernel<FIRST, COUNT, false, RZ_SHADOW_COMPAT_DEFER>), grid, block, 0, c->stream, c->dscene, c->dcamera, cfg, f, 0u);
            sort_after_shading(c, f);
            if ((c->mode_flags & HIPRZ_COMPAT_SHADOW_COLOR) && shadow_beams<COUNT>(c, f))  // coloured masks: the rays go through what they cross and collect the opacity colours
                RZ_LAUNCH((rz_shadow_packet_kernel<FIRST, COUNT, RZ_PACKET_MINW, true>), dim3(c->n_local_tiles * 4u), dim3(64), 3072, c->stream, c->dscene, c->dcamera, cfg, f);
            else if (c->mode_flags & HIPRZ_COMPAT_SHADOW_COLOR)
                RZ_LAUNCH((rz_shadow_coop_kernel<FIRST, COUNT, 3, true>), dim3(c->n_local_tiles * 4u), dim3(64), CoopLds::kBytes, c->stream, c->dscene, c->dcamera, cfg, f);
            else if (shadow_beams<COUNT>(c, f))
                RZ_LAUNCH((rz_shadow_packet_kernel<FIRST, COUNT, RZ_PACKET_MINW>), dim3(c->n_local_tiles * 4u), dim3(64), 2048, c->stream, c->dscene, c->dcamera, cfg, f);
            else
                RZ_LAUNCH((rz_shadow_coop_kernel<FIRST, COUNT, 4>), dim3(c->n_local_tiles * 4u), dim3(64), CoopLds::kBytes, c->stream, c->dscene, c->dcamera, cfg, f);
            join_sort(c);
        } else {
            RZ_LAUNCH((rz_shade_kernel<FIRST, COUNT, false, RZ_SHADOW_COMPAT>), grid, block, 0, c->stream, c->dscene, c->dcamera, cfg, f, 0u);
        }
        return;
    }
    if (!lights && c->nolight_kernels && c->n_textures == 0u) {  // no lights, no maps
        if (g.lds_scene) RZ_LAUNCH((rz_shade_kernel<FIRST, COUNT, true, RZ_SHADOW_PLAIN>), grid, block, g.blob, c->stream, c->dscene, c->dcamera, cfg, f, 0u);
        else RZ_LAUNCH((rz_shade_kernel<FIRST, COUNT, false, RZ_SHADOW_PLAIN>), grid, block, 0, c->stream, c->dscene, c->dcamera, cfg, f, 0u);
    } else if (!lights && c->nolight_kernels) {  // no next-event estimation: the instantiation without it (no shadow walk, no LDS stack)
        if (g.lds_scene) RZ_LAUNCH((rz_shade_kernel<FIRST, COUNT, true, RZ_SHADOW_NONE>), grid, block, g.blob, c->stream, c->dscene, c->dcamera, cfg, f, 0u);
        else RZ_LAUNCH((rz_shade_kernel<FIRST, COUNT, false, RZ_SHADOW_NONE>), grid, block, 0, c->stream, c->dscene, c->dcamera, cfg, f, 0u);
    } else if (g.lds_scene) {  // shadow rays inline: LDS-stack walk on the staged scene
        RZ_LAUNCH((rz_shade_kernel<FIRST, COUNT, true, 1>), grid, block, g.blob + g.stack_lds, c->stream, c->dscene, c->dcamera, cfg, f, 0u);
    } else if (lights && defer_shadows(c)) {
        // shading without shadow walks, then every shadow ray of the pass in a lean single-wave kernel
        RZ_LAUNCH((rz_shade_kernel<FIRST, COUNT, false, RZ_SHADOW_DEFER>), grid, block, 0, c->stream, c->dscene, c->dcamera, cfg, f, 0u);
        sort_after_shading(c, f);
        const dim3 sgrid(c->n_local_tiles * 4u), sblock(64);
        if (shadow_beams<COUNT>(c, f)) {
            RZ_LAUNCH((rz_shadow_packet_kernel<FIRST, COUNT, RZ_PACKET_MINW>), sgrid, sblock, 2048, c->stream, c->dscene, c->dcamera, cfg, f);
        } else if ((COUNT ? c->walk_order == 2 : c->walk_order != 0) || c->scene_tree != HIPRZ_TREE_REFERENCE) {
            RZ_LAUNCH((rz_shadow_coop_kernel<FIRST, COUNT, 4>), sgrid, sblock, CoopLds::kBytes, c->stream, c->dscene, c->dcamera, cfg, f);
        } else {
            const bool big_trees = c->trace_waves > 0 ? c->trace_waves >= 6 : c->n_nodes > kLatencyBoundNodes;
            const uint32_t top_n = std::min<uint32_t>(c->dscene.top_count, big_trees ? 170u : 272u);
            if (big_trees) RZ_LAUNCH((rz_shadow_kernel<FIRST, COUNT, 6>), sgrid, sblock, TopCache::bytes_host(top_n), c->stream, c->dscene, c->dcamera, cfg, f, top_n);
            else RZ_LAUNCH((rz_shadow_kernel<FIRST, COUNT, 4>), sgrid, sblock, TopCache::bytes_host(top_n), c->stream, c->dscene, c->dcamera, cfg, f, top_n);
        }
        join_sort(c);
    } else {  // shadow rays inline on skip links with the tree tops staged in LDS
        const uint32_t shade_top = std::min<uint32_t>(c->dscene.top_count, kTopCacheNodes);
        RZ_LAUNCH((rz_shade_kernel<FIRST, COUNT, false, 3>), grid, block, TopCache::bytes_host(shade_top), c->stream, c->dscene, c->dcamera, cfg, f, shade_top);
    }
}

}  // namespace

void launch_shade(hiprz_ctx* c, const DFrame& f, bool first, bool counted) {
    if (first) counted ? launch_shade_t<true, true>(c, f) : launch_shade_t<true, false>(c, f);
    else counted ? launch_shade_t<false, true>(c, f) : launch_shade_t<false, false>(c, f);
}

}  // namespace hiprz

#ifdef RZ_PHASE_STATS  // diagnostic build (tools/phase_stats.py): wave-level executions / active lanes of the shadow rays' wave-level walk
extern "C" int hiprz_read_shadow_phase_stats(unsigned long long out[16]) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(hiprz::rz_phase), 128);
    unsigned long long zero[16] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(hiprz::rz_phase), zero, 128);
    return 0;
}
#endif
