// hiprz_launch_trace.hip — split pipeline, first half of a pass: the closest-hit walk of every owned pixel's ray
// (traverseWorld + closestIntersection, cpu_engine_kernel.cpp:254-352) -> a 20-byte hit record per pixel.
// Instantiates rz_trace_coop_kernel / rz_trace_skip_kernel / rz_trace_kernel (hiprz_kernels.hpp).
#include "hiprz_ctx.hpp"
#include "hiprz_kernels.hpp"

namespace hiprz {
namespace {

template <bool FIRST, bool COUNT>
void launch_trace_t(hiprz_ctx* c, const DFrame& f) {
    const PassGeometry g = pass_geometry(c);
    if (c->mode_flags & kIntegratorFlags) {  // CUDA-compat integrator on the split pipeline: the cooperative walk + the medium's scattering distance
        RZ_LAUNCH((rz_trace_coop_compat_kernel<FIRST, COUNT>), dim3(c->n_local_tiles * 4u), dim3(64), CoopLds::kBytes, c->stream, c->dscene, c->dcamera, make_config(c), f);
        return;
    }
    if (g.mode == 3) {
        // one wave per workgroup: a workgroup's slot is free as soon as ITS slowest ray is done
        const dim3 grid(c->n_local_tiles * 4u), block(64);
        if ((COUNT ? c->walk_order == 2 : c->walk_order != 0) || c->scene_tree != HIPRZ_TREE_REFERENCE) {
            // front-to-back mesh walks on per-octant skip links with the cooperative triangle phase.  Register budget: 112 VGPRs are what
            // the kernel wants (4 waves per SIMD, no scratch); trees that do not live in L1 / L2 are bound by the latency of their node
            // fetches and take a fifth wave at the price of 52 B of scratch (D: 1 014 -> 964 us; C 342 -> 351, E 3 082 -> 3 279 us)
            const int waves = c->trace_waves > 0 ? c->trace_waves : (c->n_nodes > kLatencyBoundNodes ? 5 : 4);
            const bool one_leaf_world = c->dscene.n_instances != 0u && c->flat_world;  // (the general world level costs D's 5-wave build 3.5 %, C's 4-wave build 0.5 %)
            if (waves == 5 && one_leaf_world) RZ_LAUNCH((rz_trace_coop_kernel<FIRST, COUNT, 5, true>), grid, block, CoopLds::kBytes, c->stream, c->dscene, c->dcamera, f);
            else if (waves == 5) RZ_LAUNCH((rz_trace_coop_kernel<FIRST, COUNT, 5>), grid, block, CoopLds::kBytes, c->stream, c->dscene, c->dcamera, f);
            else if (waves >= 6) RZ_LAUNCH((rz_trace_coop_kernel<FIRST, COUNT, 6>), grid, block, CoopLds::kBytes, c->stream, c->dscene, c->dcamera, f);
            else if (one_leaf_world) RZ_LAUNCH((rz_trace_coop_kernel<FIRST, COUNT, 4, true>), grid, block, CoopLds::kBytes, c->stream, c->dscene, c->dcamera, f);
            else RZ_LAUNCH((rz_trace_coop_kernel<FIRST, COUNT, 4>), grid, block, CoopLds::kBytes, c->stream, c->dscene, c->dcamera, f);
        } else {
            // the reference's child order (what the work counters are anchored on), tree tops cached in LDS: 160 KiB over 24 (6 waves
            // per SIMD: big trees want occupancy) or 16 (4) single-wave workgroups per CU
            const bool big_trees = c->trace_waves > 0 ? c->trace_waves >= 6 : c->n_nodes > kLatencyBoundNodes;
            const uint32_t top_n = std::min<uint32_t>(c->dscene.top_count, big_trees ? 170u : 272u);
            if (big_trees) RZ_LAUNCH((rz_trace_skip_kernel<FIRST, COUNT, 6>), grid, block, TopCache::bytes_host(top_n), c->stream, c->dscene, c->dcamera, f, top_n);
            else RZ_LAUNCH((rz_trace_skip_kernel<FIRST, COUNT, 4>), grid, block, TopCache::bytes_host(top_n), c->stream, c->dscene, c->dcamera, f, top_n);
        }
    } else if (g.mode == 2) {
        if (g.lds_scene && c->flat_world) RZ_LAUNCH((rz_trace_kernel<FIRST, COUNT, 4, true>), g.grid, g.block, g.blob + g.walk_lds, c->stream, c->dscene, c->dcamera, f);
        else if (g.lds_scene) RZ_LAUNCH((rz_trace_kernel<FIRST, COUNT, 2, true>), g.grid, g.block, g.blob + g.walk_lds, c->stream, c->dscene, c->dcamera, f);
        else RZ_LAUNCH((rz_trace_kernel<FIRST, COUNT, 2, false>), g.grid, g.block, g.walk_lds, c->stream, c->dscene, c->dcamera, f);
    } else {
        if (g.lds_scene) RZ_LAUNCH((rz_trace_kernel<FIRST, COUNT, 1, true>), g.grid, g.block, g.blob + g.walk_lds, c->stream, c->dscene, c->dcamera, f);
        else RZ_LAUNCH((rz_trace_kernel<FIRST, COUNT, 1, false>), g.grid, g.block, g.walk_lds, c->stream, c->dscene, c->dcamera, f);
    }
}

}  // namespace

void launch_trace(hiprz_ctx* c, const DFrame& f, bool first, bool counted) {
    if (first) counted ? launch_trace_t<true, true>(c, f) : launch_trace_t<true, false>(c, f);
    else counted ? launch_trace_t<false, true>(c, f) : launch_trace_t<false, false>(c, f);
}

}  // namespace hiprz

#ifdef RZ_PHASE_STATS  // diagnostic build (tools/phase_stats.py): wave-level executions / active lanes of the walk's step kinds
extern "C" int hiprz_read_phase_stats(unsigned long long out[16]) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(hiprz::rz_phase), 128);
    unsigned long long zero[16] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(hiprz::rz_phase), zero, 128);
    return 0;
}
#endif
