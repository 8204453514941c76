// hiprz_ctx.hpp — host-side state of a context (struct hiprz_ctx) and what the translation units of libhiprz.so share:
// hiprz_api.hip (the C-ABI, scene upload, readback), hiprz_launch_trace.hip / hiprz_launch_shade.hip / hiprz_launch_batch.hip
// (the pass kernels' instantiations and their launch logic) and hiprz_sort.hip (ray reordering).
#pragma once
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <string>
#include <utility>
#include <vector>

#include "hiprz.h"
#include "hiprz_device.hpp"

namespace hiprz {

// Timer/TimeTable of the reference (engine_parts.hpp:34-74): last + EMA(0.05) per stage.
struct TimeTable {
    struct Entry {
        std::string name;
        double last_ms = 0, avg_ms = 0;
        bool seen = false;
    };
    std::vector<Entry> entries;
    void set(const char* name, double ms) {
        for (auto& e : entries)
            if (e.name == name) {
                e.last_ms = ms;
                e.avg_ms = e.seen ? e.avg_ms + (ms - e.avg_ms) * 0.05 : ms;
                e.seen = true;
                return;
            }
        entries.push_back({name, ms, ms, true});
    }
    std::string str() const {
        std::string out;
        char line[160];
        for (const auto& e : entries) {
            std::snprintf(line, sizeof line, "%-22s %9.3fms (avg %9.3fms)\n", e.name.c_str(), e.last_ms, e.avg_ms);
            out += line;
        }
        return out;
    }
};
struct StageTimer {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    double ms() const { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};

template <typename T>
struct DeviceArray {
    T* ptr = nullptr;
    size_t count = 0;
    hipError_t assign(const T* src, size_t n, hipStream_t stream) {
        if (n > count || !ptr) {
            if (ptr) (void)hipFree(ptr);
            ptr = nullptr;
            count = 0;
            hipError_t e = hipMalloc(reinterpret_cast<void**>(&ptr), sizeof(T) * (n ? n : 1));
            if (e != hipSuccess) return e;
            count = n ? n : 1;
        }
        if (n) return hipMemcpyAsync(ptr, src, sizeof(T) * n, hipMemcpyHostToDevice, stream);
        return hipSuccess;
    }
    hipError_t resize(size_t n) {
        if (n <= count && ptr) return hipSuccess;
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        count = 0;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&ptr), sizeof(T) * (n ? n : 1));
        if (e == hipSuccess) count = n ? n : 1;
        return e;
    }
    void release() {
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        count = 0;
    }
};

}  // namespace hiprz

namespace hiprz {
// one mesh of a scene whose trees are built on the device (hiprz_build.hip)
struct DeviceMesh {
    uint32_t tri_first = 0, n_tris = 0;  // its triangles in the device order
    uint32_t ref_first = 0;              // ... and in the uploaded snapshot's order (a mesh's triangles are contiguous in both)
    uint32_t region = 0xFFFFFFFFu;       // slot of its root in the node arrays (RZ_END: too small to build, stays one leaf)
    uint32_t leaf_slot = 0xFFFFFFFFu;    // ... the slot of that one leaf (the uploaded placeholder), whose box a refit fits again
    uint32_t n_slots = 0;                // nodes emitted
    float bb_min[3] = {0, 0, 0}, bb_max[3] = {0, 0, 0};
};
}  // namespace hiprz

// What belongs to ONE camera of the world: its record, the per-pixel path state and accumulators, the device-resident pass index, the
// ray-order and shadow hand-over buffers sized for its resolution, the graph that was captured over these pointers.  The reference
// renders every enabled camera per call (cpu_engine_renderer.cpp:97-117); a context keeps one of these per camera and the calls
// address the selected one (hiprz_select_camera): hiprz_ctx IS-A frame state, the others are parked.
// the HIPRZ_COMPAT_* flags that change the integration (everything but the reprojection of history at a restart)
constexpr uint32_t kIntegratorFlags = HIPRZ_MODE_CUDA_COMPAT & ~HIPRZ_COMPAT_REPROJECTION;

struct hiprz_frame_state {
    hiprz_camera camera{};
    hiprz::DCamera dcamera{};
    bool have_camera = false;
    uint32_t tiles_x = 0, tiles_y = 0, n_local_tiles = 0;
    uint64_t owned_pixels = 0;
    hiprz::DeviceArray<float4> st0, st1, accum, hit0;
    hiprz::DeviceArray<uint32_t> hit1;
    bool rgba8_valid = false;  // the resident kernel tone-maps on its way out: hiprz_tonemap has nothing to do
    hiprz::DeviceArray<float2> st2;
    hiprz::DeviceArray<float> depth;
    hiprz::DeviceArray<uint32_t> rgba8;
    hiprz::DeviceArray<float4> image_f4;  // row-major staging for readback
    hiprz::DeviceArray<uint32_t> state_md;
    hiprz::DeviceArray<float> state_ray;
    hiprz::DeviceArray<uint32_t> pass_dev;
    bool reset_pending = true;
    uint32_t passes = 0;
    uint64_t ray_count = 0;
    // hipGraph of one batch of cumulative passes ([pass kernel, pass update] x n): replayed while nothing that
    // the captured kernel arguments depend on has changed (scene, camera, config, shard, variants)
    hipGraphExec_t graph_exec = nullptr;
    uint32_t graph_passes = 0;
    bool graph_valid = false;
    // every byte the captured launches were given (kernel arguments, grid sizes, the settings that pick a kernel instantiation): a
    // graph is replayed only while an eager launch would pass exactly the same — whatever invalidation a code path may have missed
    std::vector<unsigned char> graph_key;
    // ray reordering between passes (split pipeline, hiprz_sort.hip): keys from the shade kernel -> radix sort -> the permutation the
    // next trace kernel follows
    hiprz::DeviceArray<uint32_t> sort_keys, sort_perm;
    struct SortTemp {  // ping-pong buffers of one sort; the ray sort and the shadow-ray sort of a pass run side by side, each on its own set
        hiprz::DeviceArray<uint32_t> keys_out, vals_a, vals_b, counts, row_total;
    } sort_temp[2];
    bool sort_beside = false;  // a sort is running on the auxiliary stream: join_sort() before its order is used
    bool perm_valid = false;  // sort_perm holds the order of the NEXT cumulative pass's rays
    hiprz::DeviceArray<uint32_t> shadow_keys, shadow_perm;  // deferred shadow rays follow their own order (hiprz_device.hpp: DFrame::shadow_key)
    bool sorted_this_pass = false;  // the deferred shadow kernel wants the NEXT pass's ray order: the sort then runs before it
    hiprz::DeviceArray<float4> nee;  // DFrame::nee
    // resident kernels, heaviest first (DFrame::launch_order): what every unit (a tile of rz_batch_kernel, a wave of rz_wave_batch_kernel)
    // cost in the last batch, the units by falling cost, the workspace of the sort that orders them
    hiprz::DeviceArray<uint32_t> unit_cost, launch_order, order_keys;
    SortTemp order_sort;
    uint32_t order_units = 0;           // how many units launch_order permutes (0: none yet)
    uint32_t batches_since_order = 0;   // resident batches since the order was last derived
    // HIPRZ_COMPAT_REPROJECTION: the frame a restart replaces (accumulator, first-hit depth, the camera it was rendered from)
    hiprz::DeviceArray<float4> prev_accum;
    hiprz::DeviceArray<float> prev_depth;
    hiprz_camera frame_camera{};  // camera of the frame being accumulated
    bool frame_started = false;   // a first pass ran since the frame buffers were (re)allocated
    bool history_ready = false;   // prev_accum / prev_depth already hold the whole previous frame (a multi-device head assembled it)
    float temporal_blend = 0.75f;
    hiprz::DeviceArray<uint8_t> gather;  // multi-device head: the peers' tile buffers land here before one launch untiles them all
    hiprz::DeviceArray<float4> sum_accum;  // HIPRZ_SHARD_SAMPLES head: the parts' accumulators summed, tile-major like `accum`
};

struct hiprz_ctx : hiprz_frame_state {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string error;
    hiprz::TimeTable timings;
    // second stream + events: the sort of the NEXT pass's rays runs beside the shadow-ray kernel of this pass (hiprz_launch_shade.hip)
    hipStream_t aux_stream = nullptr;
    hipEvent_t aux_fork = nullptr, aux_join = nullptr;

    // multi-device (hiprz_create_multi): the head context owns shard 0 and one peer context per further device; every call fans out,
    // readbacks gather the peers' tiles over P2P copies.  Peers have no peers.
    std::vector<hiprz_ctx*> peers;
    hipEvent_t peer_done = nullptr;  // peer side: recorded on its stream when its tiles are ready, awaited by the head's stream
    hipEvent_t history_done = nullptr;  // head side: the assembled history of a restarted frame has reached every peer
    uint32_t user_rank = 0, user_world = 1;  // hiprz_set_shard as the caller sees it; peers refine it: (rank * n + r, world * n)
    uint32_t shard_mode = 0;          // HIPRZ_SHARD_TILES | HIPRZ_SHARD_SAMPLES (head): how the parts divide the context's share
    hipEvent_t sum_done = nullptr;    // head, sample mode: the last sum of the parts has read the staging slices (the peers' next copies wait for it)
    bool sum_recorded = false;

    // cameras (hiprz_set_camera_count / hiprz_select_camera)
    std::vector<hiprz_frame_state> parked;  // slot [active_camera] is empty while that camera's state lives in the context itself
    uint32_t active_camera = 0;

    // scene mirror
    hiprz::DeviceArray<uint8_t> hot;  // nodes | tlas_order | instances | tris | tri_attrs | materials | inst_materials
    hiprz::DeviceArray<uint32_t> node_skip;
    hiprz::DeviceArray<uint32_t> nodes64;
    int walk_order = 1;  // 0 = meshes in the reference's child order, 1 = front-to-back, 2 = also when counting (hiprz_set_walk_order)
    hiprz::DeviceArray<hiprz_texture> textures;
    hiprz::DeviceArray<uint8_t> texels;
    hiprz::DeviceArray<hiprz_spot_light> spot_lights;
    hiprz::DeviceArray<hiprz_direct_light> direct_lights;
    hiprz::DScene dscene{};
    bool have_scene = false;
    bool scene_shared = false;  // a second stream on the head's device: `dscene` points into the head context's buffers
    uint32_t stack_entries = 2;  // LDS stack entries per lane the trees need (MODE 1)
    bool lds_scene = false;      // hot blob is staged into LDS by every workgroup
    int lds_scene_override = -1; // -1 auto, 0 never, 1 always (if it fits at all)

    uint32_t rank = 0, world = 1;  // the shard this context renders
    // 0 fused (one kernel per pass), 1 split (trace kernel -> shade kernel per pass), 2 resident (one kernel per batch
    // of passes).  -1: resident when the scene is staged in LDS (config B: as fast as split on a whole frame, 2.26 ms per
    // 8 passes, and 0.34 vs 0.45 ms on an eighth of it — per-pass launch/ramp/tail costs vanish), else split (10-20 % faster
    // than fused on configs C, D; the resident kernel has no LDS room for the tree-top cache).
    int pipeline_setting = -1;
    int pipeline = 1;  // resolved by resolve_pipeline() at upload / set time and before a render call
    uint32_t wave_resident_max = 1u << 30;  // HIPRZ_WAVE_RESIDENT_MAX: scenes without lights that are not staged in LDS run the resident pipeline
                                         // (rz_wave_batch_kernel) while a shard has at most this many waves — no limit since the end of round 3.
                                         // Measured on MI355X, split / resident, ms per step of 8 passes: an eighth of a 1080p frame C 1.18 /
                                         // 0.61, D 3.78 / 2.69; half C 2.46 / 1.86, D 5.79 / 3.91; a whole frame (32 400 waves), since the walk's
                                         // instance level: C 3.69 / 3.38, D 7.28 / 7.19 (it was C 3.78 / 3.82 before); a 4K frame (129 600 waves):
                                         // C 14.26 / 12.73, D 27.26 / 27.18.  (The kernel keeps the register budget of 4 waves per SIMD: with 5 — what
                                         // D's trace kernel likes — the shading spills: D 7.18 -> 7.51, C 3.38 -> 3.81 ms per step.)
    // device-built trees (hiprz_set_tree(HIPRZ_TREE_DEVICE), hiprz_build.hip): the 32-byte node records of the whole scene in a buffer of
    // their own (the hot blob's node section only holds the uploaded prefix), the workspaces of build and refit, the meshes
    hiprz::DeviceArray<uint8_t> dev_nodes, has_mesh, build_temp;
    hiprz::DeviceArray<uint32_t> slot_parent, ref_to_dev, refit_visit, world_items;
    hiprz::DeviceArray<hiprz_tri> update_tris;
    hiprz::DeviceArray<hiprz_tri_attr> update_attrs;
    hiprz_frame_state::SortTemp build_sort;
    hiprz::DeviceArray<uint32_t> shadow_nodes64, shadow_order;  // the shadow rays' own world tree (build_shadow_world_tree)
    std::vector<uint32_t> world_members;                         // the instances of the world tree (those with a mesh), by rising id
    std::vector<hiprz::DeviceMesh> device_meshes;
    std::vector<uint32_t> instance_mesh;          // instance -> index into device_meshes (RZ_END: no mesh)
    std::vector<hiprz_instance> device_instances; // the instance records as the device holds them (hiprz_update_instances keeps what it does not replace)
    uint32_t node_capacity = 0, n_tris = 0, n_tlas_order = 0, world_region = 0, world_slots = 0;
    hiprz::DeviceArray<unsigned long long> counters_dev;
    hiprz::DeviceArray<int32_t> pick_dev;

    hiprz_config config{8u, 8u, 1u, 1u, 20240501u};
    int traversal_mode = -1;  // -1 = choose per scene (effective_mode)
    uint32_t tree_mode = 0;   // HIPRZ_TREE_* (hiprz_set_tree), applied by hiprz_upload_scene
    uint32_t scene_tree = 0;  // ... of the scene that is uploaded now
    bool device_sah = false;  // HIPRZ_TREE_DEVICE_SAH: the device's mesh trees by the binned surface-area build instead of Morton order
    bool build_sah = false;   // ... of the scene that is uploaded now (HIPRZ_TREE_AUTO decides per scene)
    uint32_t mode_flags = 0;  // HIPRZ_COMPAT_* (hiprz_set_mode): an integrator flag routes every pass through rz_compat_pass_kernel
    uint32_t graph_captures = 0;  // how often a batch was captured + instantiated (hiprz_graph_captures)
    uint32_t n_textures = 0;  // of the uploaded scene
    int batch_waves = 0;  // HIPRZ_BATCH_WAVES=4: never the 5-wave build of the plain batch kernel
    int nolight_kernels = 1;  // scenes without lights use the instantiations without next-event estimation (HIPRZ_NOLIGHT_KERNELS=0: the general ones)
    bool flat_world = false;  // the uploaded world tree is one leaf of at most 8 instances: the binned walk tests their boxes up front (MODE 4)
    int sort_bits = 0;    // most significant key bits the radix sorts look at; 0 = by frame size (HIPRZ_SORT_BITS)
    bool shadow_tree = true;  // the wave-level shadow walk takes a world tree of its own (build_shadow_world_tree); HIPRZ_SHADOW_TREE=0: the reference's
    int shadow_packet = -1;  // sorted shadow rays walked by the wave (rz_shadow_packet_kernel): -1 where the beams are narrow enough (launch_shade), 0 never, 1 always (HIPRZ_SHADOW_PACKET)
    int shadow_sort = 1;  // HIPRZ_SHADOW_SORT=0: the shadow kernel follows the next pass's ray order instead
    int sort_rays = -1;  // -1 auto (on for scenes walked with MODE 3), 0 off, 1 on
    bool defer_shadow_rays = true;  // HIPRZ_DEFER_SHADOWS=0: walk them inside the shade kernel
    int heavy_first = 1;            // HIPRZ_HEAVY_FIRST=0: resident kernels launch their units in unit order (rounds 1 - 3)
    int trace_waves = 0;  // 0 = by tree size; HIPRZ_TRACE_WAVES = 4 | 6 forces the register budget
    uint32_t n_nodes = 0;
    bool time_kernels = false;  // record events around the trace and the shade kernel of every pass of a batch
    std::vector<hipEvent_t> kernel_events;
    uint32_t kernel_event_passes = 0;
    bool use_graph = true;
    bool xcd_swizzle = false;  // measured: banding the image per XCD concentrates the expensive region on few XCDs (D: 4.3 -> 5.1 ms)

    // kernel timing (hip events on `stream` around each render batch)
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending_events;
    std::vector<uint32_t> pending_launches;
    std::vector<hipEvent_t> event_pool;
};

namespace hiprz {

int fail(hiprz_ctx* ctx, int code, const std::string& msg);

// Every kernel instantiation a launcher can select registers its host stub at load time (RZ_LAUNCH below); hiprz_create resolves each of
// them against the loaded code objects once per device (hipFuncGetAttributes) and refuses to come up, naming the kernel, when one is
// missing — a launch of such a kernel ends the process inside the HIP runtime ("Cannot find Symbol with name ...", round 3), which no
// return code can report.  tools/check_kernels.py proves the same for the files at build time.
struct KernelEntry {
    const void* stub;
    const char* name;  // __PRETTY_FUNCTION__ of the launch site's registration: holds the instantiation's name
};
void register_kernel(const void* stub, const char* name);
template <auto Kernel>
struct LaunchSite {
    static const char* name() { return __PRETTY_FUNCTION__; }
    static const bool registered;
};
template <auto Kernel>
const bool LaunchSite<Kernel>::registered = (register_kernel(reinterpret_cast<const void*>(Kernel), LaunchSite<Kernel>::name()), true);
#define RZ_LAUNCH(kernel, ...)                                  \
    do {                                                        \
        (void)hiprz::LaunchSite<&(kernel)>::registered;         \
        hipLaunchKernelGGL(kernel, __VA_ARGS__);                \
    } while (0)

#define RZ_HIP(ctx, call)                                                                                       \
    do {                                                                                                        \
        hipError_t rz_e = (call);                                                                               \
        if (rz_e != hipSuccess)                                                                                 \
            return hiprz::fail(ctx, HIPRZ_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(rz_e));    \
    } while (0)

constexpr uint32_t kLatencyBoundNodes = 32768u;  // trees beyond ~1 MiB of nodes: fetches come from L2 / HBM, occupancy hides them
constexpr uint32_t kTopCacheNodes = 682u;        // 682 x 36 B = 24 KiB per workgroup: ~9 levels of every tree, 5 workgroups per CU
constexpr size_t kLdsSceneLimit = 52u * 1024u;   // per workgroup: 3 x 52 KiB < 160 KiB per CU

// choices derived from the context's settings and the uploaded scene (hiprz_api.hip)
int effective_mode(const hiprz_ctx* c);
bool defer_shadows(const hiprz_ctx* c);
bool use_lds_scene(const hiprz_ctx* c);
bool sort_enabled(const hiprz_ctx* c);
bool wave_resident(const hiprz_ctx* c);  // resident pipeline on a scene that is not staged in LDS: rz_wave_batch_kernel
int effective_sort_bits(const hiprz_ctx* c);
DConfig make_config(const hiprz_ctx* c);
// launch geometry of the 256-thread pass kernels: one workgroup per owned 32x8 tile
struct PassGeometry {
    dim3 grid, block;
    bool lds_scene;    // the hot blob is staged into LDS by every workgroup
    size_t blob;       // its bytes (0 when not staged)
    int mode;          // walk of this launch: 1 LDS stack, 2 workgroup-binned, 3 skip links (split pipeline, global scene)
    size_t stack_lds;  // LDS stack columns of the MODE 1 walk (and of inline shadow rays)
    size_t walk_lds;   // workspace of the closest-hit walk
};
PassGeometry pass_geometry(const hiprz_ctx* c);

// launch units.  `first`: renderFirstPass instead of renderCumulativePass; `counted`: the instrumented instantiation.
void launch_trace(hiprz_ctx* c, const DFrame& f, bool first, bool counted);   // split pipeline: closest-hit walk -> hit records
void launch_shade(hiprz_ctx* c, const DFrame& f, bool first, bool counted);   // split pipeline: shading (+ deferred shadow rays and their sorts)
void launch_fused(hiprz_ctx* c, const DFrame& f, bool first, bool counted);   // fused pipeline: one kernel per pass
void launch_batch(hiprz_ctx* c, const DFrame& f, uint32_t n_passes, bool counted, hipEvent_t before, hipEvent_t after);  // resident pipeline
void launch_sort(hiprz_ctx* c, bool beside = false);  // keys of the next rays -> the permutation the next trace kernel follows;
                                                      // beside: on the auxiliary stream, joined by join_sort()
void join_sort(hiprz_ctx* c);
void launch_shadow_sort(hiprz_ctx* c);  // keys of the pass's shadow rays -> the order the shadow kernel follows
void launch_sort_identity(hiprz_ctx* c);  // the identity order (no sort has run on this frame's rays yet)
// device-side tree build and refit (hiprz_build.hip)
uint32_t device_build_regions(std::vector<DeviceMesh>& meshes, uint32_t first_free_slot);
int device_build_mesh_trees(hiprz_ctx* c, std::vector<DeviceMesh>& meshes, const std::vector<uint32_t>& instance_mesh, bool validate);
int device_build_world_tree(hiprz_ctx* c, bool validate);
int device_update_triangles(hiprz_ctx* c, uint32_t first, uint32_t n, const hiprz_tri* tris, const hiprz_tri_attr* attrs);
int sort_workspace(hiprz_ctx* c, size_t n);  // (re)allocates the sort's buffers for n keys
int sort_temp_resize(hiprz_ctx* c, hiprz_frame_state::SortTemp& t, size_t n);
void sort_u32(hipStream_t stream, uint32_t* keys, uint32_t n, int key_bits, uint32_t* perm, uint32_t* sorted_keys, hiprz_frame_state::SortTemp& t);  // keys destroyed

}  // namespace hiprz
