// hiprz_api.hip — the device half of the C-ABI declared in include/hiprz.h: context, scene mirroring, render calls, readback.
//
// Replaces, for the HIPGPU backend, what the reference's CUDA backend does in cuda_engine_core.cu (host<->device
// mirroring, readback), cuda_engine_renderer.cu (launch sequence) and cuda_postprocess_kernel.cu (tone map, pass update).
// The pass kernels live in hiprz_kernels.hpp and are instantiated by hiprz_launch_*.hip.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "hiprz.h"
#include "hiprz_ctx.hpp"
#include "hiprz_device.hpp"

using namespace hiprz;

// =======================================================================================
// Small kernels: pass index, tone map, tile <-> image, picking, self-test
// =======================================================================================

// resident kernels, heaviest first: sort keys from the units' measured costs (falling cost = rising key)
__global__ void __launch_bounds__(256) rz_order_keys_kernel(const uint32_t* cost, uint32_t* keys, uint32_t n) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) keys[i] = 0x00FFFFFFu - (cost[i] < 0x00FFFFFFu ? cost[i] : 0x00FFFFFFu);
}

// passUpdate / segmentUpdate (cuda_postprocess_kernel.cu:95-104, cuda_render_kernel.cu:122-129):
// the pass index lives on the device so a captured graph replays without new arguments.
__global__ void rz_pass_update_kernel(uint32_t* pass) { *pass += 1u; }
__global__ void rz_pass_reset_kernel(uint32_t* pass) { *pass = 0u; }
__global__ void rz_pass_add_kernel(uint32_t* pass, uint32_t n) { *pass += n; }

// toneMap (cuda_postprocess_kernel.cu:38-93; CPU: cpu_engine_renderer.cpp:224-235)
__global__ void __launch_bounds__(256) rz_tonemap_tiles_kernel(const float4* accum, uint32_t* rgba8, uint32_t n, float aperture,
                                                               float exposure_time) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float4 a = accum[i];
    rgba8[i] = tonemap(col4{a.x, a.y, a.z, a.w}, aperture, exposure_time);
}
__global__ void __launch_bounds__(256) rz_tonemap_image_kernel(const float4* image, uint32_t* rgba8, uint32_t n, float aperture,
                                                               float exposure_time) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float4 a = image[i];
    rgba8[i] = tonemap(col4{a.x, a.y, a.z, a.w}, aperture, exposure_time);
}

// HIPRZ_SHARD_SAMPLES: the parts of a context rendered the same pixels on different seed streams; what leaves the context is the sum of
// their accumulators (colour sums and finished-path counts), taken in part order — own + staged[0] + staged[1] + ... — so that the
// result does not depend on when a part finished
__global__ void __launch_bounds__(256) rz_sum_parts_kernel(const float4* own, const float4* staged, size_t stride, uint32_t n_staged, float4* out, uint32_t n) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    float4 a = own[i];
    for (uint32_t r = 0; r < n_staged; ++r) {
        const float4 b = staged[size_t(r) * stride + i];
        a.x += b.x, a.y += b.y, a.z += b.z, a.w += b.w;
    }
    out[i] = a;
}

// tile-major (owned tiles of shard rank/world) -> row-major full frame
template <typename T>
__global__ void __launch_bounds__(256) rz_untile_kernel(const T* tiles, T* image, uint32_t width, uint32_t height,
                                                        uint32_t tiles_x, uint32_t rank, uint32_t world) {
    uint32_t tx, ty;
    shard_tile(blockIdx.x, tiles_x, rank, world, tx, ty);
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t x = tx * 32u + wave * 8u + (lane & 7u), y = ty * 8u + (lane >> 3);
    if (x < width && y < height) image[size_t(y) * width + x] = tiles[size_t(blockIdx.x) * 256u + threadIdx.x];
}
// the gathered tiles of ALL shards (shard r at tiles + r * part_stride elements) -> row-major full frame, one launch
// (slice blockIdx.y holds shard rank0 + blockIdx.y of `world`)
template <typename T>
__global__ void __launch_bounds__(256) rz_untile_gathered_kernel(const T* tiles, size_t part_stride, T* image, uint32_t width, uint32_t height,
                                                                 uint32_t tiles_x, uint32_t n_tiles, uint32_t world, uint32_t rank0) {
    const uint32_t part = blockIdx.y;
    if (blockIdx.x * world + rank0 + part >= n_tiles) return;  // the higher shards own one tile less
    uint32_t tx, ty;
    shard_tile(blockIdx.x, tiles_x, rank0 + part, world, tx, ty);
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t x = tx * 32u + wave * 8u + (lane & 7u), y = ty * 8u + (lane >> 3);
    if (x < width && y < height) image[size_t(y) * width + x] = tiles[part * part_stride + size_t(blockIdx.x) * 256u + threadIdx.x];
}
__global__ void __launch_bounds__(256) rz_untile_state_kernel(const float4* st0, const float4* st1, const float2* st2, float* ray9,
                                                              uint32_t* md2, uint32_t width, uint32_t height, uint32_t tiles_x,
                                                              uint32_t rank, uint32_t world) {
    uint32_t tx, ty;
    shard_tile(blockIdx.x, tiles_x, rank, world, tx, ty);
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t x = tx * 32u + wave * 8u + (lane & 7u), y = ty * 8u + (lane >> 3);
    if (x >= width || y >= height) return;
    const size_t i = size_t(blockIdx.x) * 256u + threadIdx.x, o = size_t(y) * width + x;
    const float4 a = st0[i], b = st1[i];
    const float2 c = st2[i];
    float* r = ray9 + 9 * o;
    r[0] = a.x, r[1] = a.y, r[2] = a.z, r[3] = a.w, r[4] = b.x, r[5] = b.y, r[6] = b.z, r[7] = b.w, r[8] = c.x;
    const uint32_t bits = __float_as_uint(c.y);
    md2[2 * o] = bits & 0xFFFFu;
    md2[2 * o + 1] = (bits >> 16) & 0xFFu;
}

// Kernel::rayCast (cpu_engine_kernel.cpp:102-111, 483-501): one thread.
__global__ void rz_pick_kernel(const DScene s, const DCamera cam, uint32_t x, uint32_t y, float depth, int32_t* out4) {
    Ray ray;
    generate_simple_ray(cam, ray, x, y);
    ray.near_ = depth * 0.99f;
    ray.far_ = depth * 1.01f;
    Hit hit;
    hit.instance = -1, hit.triangle = 0u, hit.bx = hit.by = 0.0f, hit.external = true;
    Counters cnt;
    out4[0] = out4[1] = out4[2] = -1, out4[3] = 0;
    if (s.n_instances != 0u && closest_hit_skip<false, false, true>(s, TopCache{nullptr, nullptr, 0u}, ray, hit, cnt) == 2) {
        const uint32_t inst = uint32_t(hit.instance);
        const uint32_t material_base = __float_as_uint(s.instances[7 * inst + 1].w);
        const uint32_t material_count = __float_as_uint(s.instances[7 * inst + 2].w);
        uint32_t slot = __float_as_uint(s.tris[3 * hit.triangle].w) & HIPRZ_TRI_MATERIAL_MASK;
        if (slot > 63u) slot = 63u;
        out4[0] = hit.instance;
        out4[1] = int32_t(slot);
        out4[2] = slot < material_count ? s.inst_materials[material_base + slot] : -1;
        out4[3] = int32_t(__float_as_uint(s.tris[3 * hit.triangle + 1].w));  // hiprz_tri::source_index
    }
}

// Device self-test: div_shared() must equal the correctly rounded `/` bit for bit over its
// whole stated operand range.  out[0] = mismatches, out[1] = cases tested.
__global__ void __launch_bounds__(256) rz_selftest_div_kernel(uint32_t n_per_thread, uint32_t seed, unsigned long long* out) {
    uint32_t h = mix32(seed ^ (blockIdx.x * 256u + threadIdx.x) * 0x9E3779B9u);
    uint32_t bad = 0, tested = 0, undecided = 0;
    for (uint32_t i = 0; i < n_per_thread; ++i) {
        h = mix32(h + i);
        // d: random sign/mantissa, exponent in [-40, 2); n: zero or exponent in [-84, 41)
        const uint32_t de = 127u - 40u + (h >> 8) % 42u;
        const float d = __uint_as_float((h & 0x80000000u) | (de << 23) | (mix32(h) & 0x7FFFFFu));
        const uint32_t g = mix32(h ^ 0xA5A5A5A5u);
        const uint32_t ne = 127u - 84u + (g >> 8) % 125u;
        float n = __uint_as_float((g & 0x80000000u) | (ne << 23) | (mix32(g) & 0x7FFFFFu));
        if ((g & 0xFFu) == 0u) n = 0.0f;
        const float y = refined_rcp(d);
        const float n2 = n == 0.0f ? 0.0f : __uint_as_float(__float_as_uint(n) ^ (mix32(g + 7u) & 0x007FFFFFu));  // second numerator, same exponent
        const f2 fast = div_shared2(f2{n, n2}, d, y);
        const float exact = n / d, exact2 = n2 / d;
        tested += 2;
        if (__float_as_uint(fast.x) != __float_as_uint(exact) && !(fast.x == 0.0f && exact == 0.0f)) bad += 1;
        if (__float_as_uint(fast.y) != __float_as_uint(exact2) && !(fast.y == 0.0f && exact2 == 0.0f)) bad += 1;
        if (__float_as_uint(div_shared(n, d, y)) != __float_as_uint(fast.x)) bad += 1;
        // sincosf must return what sinf and cosf return: the samplers' angles are in [0, 2*pi], test a wider range
        const float angle = __uint_as_float((g & 0x80000000u) | ((118u + (h >> 11) % 16u) << 23) | (mix32(h + 3u) & 0x7FFFFFu));
        float sn, cs;
        sincosf(angle, &sn, &cs);
        if (__float_as_uint(sn) != __float_as_uint(sinf(angle)) || __float_as_uint(cs) != __float_as_uint(cosf(angle))) bad += 1;
        // the filtered box test (box_filter): whenever it claims to know, the exact test must agree.  A ray from a random point towards a
        // random direction against a random box around the origin region — every second case with a face of the box exactly ON the ray's
        // range end or through the ray's origin plane, where comparisons are decided by a rounding
        {
            auto unit = [&](uint32_t k) { return float(mix32(h + 0x9E37u * k) >> 8) * (1.0f / 16777216.0f); };
            WalkRay r;
            r.o = V3(unit(1) * 8.0f - 4.0f, unit(2) * 8.0f - 4.0f, unit(3) * 8.0f - 4.0f);
            v3 dir = V3(unit(4) * 2.0f - 1.0f, unit(5) * 2.0f - 1.0f, unit(6) * 2.0f - 1.0f);
            if ((h & 7u) == 0u) dir.x *= 1.0e-6f;  // nearly parallel to two faces
            dir = normalized(dir);
            r.d = dir, r.near_ = 0.0f, r.far_ = (h & 1u) ? unit(7) * 6.0f : RZ_FLT_MAX;
            prepare<true>(r, true);
            const float cx = unit(8) * 6.0f - 3.0f, cy = unit(9) * 6.0f - 3.0f, cz = unit(10) * 6.0f - 3.0f;
            float hx = unit(11) * 2.0f, hy = unit(12) * 2.0f, hz = unit(13) * 2.0f;
            if ((h & 48u) == 0u) hz = 0.0f;  // a flat box (an axis-aligned wall)
            float4 b0 = make_float4(cx - hx, cx + hx, cy - hy, cy + hy), b1 = make_float4(cz - hz, cz + hz, 0.0f, 0.0f);
            if (h & 2u) {  // put the far end of the range exactly where the ray crosses the box's lower x face (when it points that way)
                const float t = (b0.x - r.o.x) / r.d.x;
                if (t > 0.0f && t < 1.0e6f) r.far_ = t;
            }
            if (r.fast) {
                bool missed, hit;
                box_filter(b0, b1, r, missed, hit);
                const bool exact = box_hit_unpacked<false>(b0, b1, r);
                tested += 1;
                if ((hit && !exact) || (missed && exact) || (hit && missed)) bad += 1;
                if (!hit && !missed) undecided += 1;
            }
        }
    }
    atomicAdd(&out[0], (unsigned long long)bad);
    atomicAdd(&out[1], (unsigned long long)tested);
    atomicAdd(&out[2], (unsigned long long)undecided);
}

// =======================================================================================
// Host side of the context
// =======================================================================================
namespace hiprz {
thread_local std::string g_create_error;
int fail(hiprz_ctx* ctx, int code, const std::string& msg) {
    if (ctx) ctx->error = msg;
    else g_create_error = msg;
    return code;
}
// the launch sites' kernel stubs (hiprz_ctx.hpp: RZ_LAUNCH); filled during static initialisation of the translation units
std::vector<KernelEntry>& kernel_table() {
    static std::vector<KernelEntry> table;
    return table;
}
void register_kernel(const void* stub, const char* name) { kernel_table().push_back(KernelEntry{stub, name}); }
// Every kernel a launcher can select must resolve in the code objects this process loaded, on the device a context is created for:
// checked once per device, at the first hiprz_create.  (What it costs: the runtime loads every code object of the library up front
// instead of at the first launch from it.)
int resolve_kernels(int device) {
    static std::vector<int> checked;
    for (int d : checked)
        if (d == device) return HIPRZ_OK;
    for (const KernelEntry& e : kernel_table()) {
        hipFuncAttributes attr;
        const hipError_t err = hipFuncGetAttributes(&attr, e.stub);
        if (err != hipSuccess) {
            (void)hipGetLastError();
            return fail(nullptr, HIPRZ_ERR_DEVICE, std::string("libhiprz.so is inconsistent: the loaded gfx950 code objects do not hold a kernel the host side can launch (") +
                                                       hipGetErrorString(err) + "): " + e.name + " — rebuild (make -C rayzath_amd/csrc; tools/check_kernels.py)");
        }
    }
    checked.push_back(device);
    return HIPRZ_OK;
}
// settings are shared by the cameras of a context: a change invalidates the graph of every one of them
void invalidate_graphs(hiprz_ctx* c) {
    c->graph_valid = false;
    for (auto& f : c->parked) f.graph_valid = false;
}
}  // namespace hiprz

// Multi-device contexts (hiprz_create_multi): a call on the head is repeated on every peer first; a peer's failure is the call's.
#define RZ_FANOUT(c, call)                                                                                                     \
    for (hiprz_ctx* p : (c)->peers) {                                                                                         \
        const int rz_rc = (call);                                                                                             \
        if (rz_rc != HIPRZ_OK) return fail(c, rz_rc, "device " + std::to_string(p->device) + ": " + p->error);              \
    }

// Scene calls: a peer on ANOTHER device mirrors the scene itself; a peer on the head's own device (a second stream on the same GPU)
// walks the head's copy — one scene blob per device, however many streams share it.
#define RZ_FANOUT_OTHER_DEVICES(c, call)                                                                                       \
    for (hiprz_ctx* p : (c)->peers) {                                                                                         \
        if (p->device == (c)->device) continue;                                                                               \
        const int rz_rc = (call);                                                                                             \
        if (rz_rc != HIPRZ_OK) return fail(c, rz_rc, "device " + std::to_string(p->device) + ": " + p->error);              \
    }

namespace {

// a same-device peer takes over the head's view of the scene (pointers into the head's buffers, every derived figure)
void adopt_scene(hiprz_ctx* p, const hiprz_ctx* head) {
    p->dscene = head->dscene, p->have_scene = head->have_scene, p->stack_entries = head->stack_entries, p->lds_scene = head->lds_scene;
    p->n_nodes = head->n_nodes, p->flat_world = head->flat_world, p->n_textures = head->n_textures, p->scene_tree = head->scene_tree;
    p->tree_mode = head->tree_mode, p->device_sah = head->device_sah, p->build_sah = head->build_sah, p->n_tris = head->n_tris, p->n_tlas_order = head->n_tlas_order;
    p->scene_shared = true;
    invalidate_graphs(p);
    p->reset_pending = true;
    for (auto& f : p->parked) f.reset_pending = true;
}
void share_scene_with_streams(hiprz_ctx* c) {
    for (hiprz_ctx* p : c->peers)
        if (p->device == c->device) adopt_scene(p, c);
}

struct TreeCheck {
    const hiprz_scene* sc;
    std::vector<uint32_t>& skip;
    std::vector<uint8_t> visited;
    uint32_t max_depth = 0;
    std::string error;
    std::vector<uint32_t> world_leaves;

    // Walks one tree from `root`, verifies every index it will make the kernel follow, fills
    // the skip links, returns false on the first violation.  `is_world`: leaves index tlas_order.
    bool walk(uint32_t root, bool is_world) {
        struct Item {
            uint32_t node, skip, depth;
        };
        std::vector<Item> stack{{root, RZ_END, 1u}};
        while (!stack.empty()) {
            const Item it = stack.back();
            stack.pop_back();
            if (it.node >= sc->n_nodes) return err("node index out of range");
            if (visited[it.node]) return err("node reachable twice (trees must be disjoint and acyclic)");
            visited[it.node] = 1;
            skip[it.node] = it.skip;
            if (it.depth > max_depth) max_depth = it.depth;
            if (it.depth > 64u) return err("tree deeper than 64 levels");
            const hiprz_node& n = sc->nodes[it.node];
            if (n.meta & HIPRZ_NODE_LEAF) {
                const uint64_t end = uint64_t(n.begin) + (n.meta & HIPRZ_NODE_COUNT_MASK);
                if (end > (is_world ? sc->n_tlas_order : sc->n_tris)) return err("leaf range out of bounds");
                if (is_world) world_leaves.push_back(it.node);
            } else {
                if (uint64_t(n.begin) + 1 >= sc->n_nodes) return err("child index out of range");
                stack.push_back({n.begin + 1, it.skip, it.depth + 1});
                stack.push_back({n.begin, n.begin + 1, it.depth + 1});
            }
        }
        return true;
    }
    bool err(const char* m) {
        error = m;
        return false;
    }
};

void release_frame(hiprz_frame_state* c) {
    c->st0.release(), c->st1.release(), c->st2.release(), c->accum.release(), c->depth.release(), c->rgba8.release();
    c->hit0.release(), c->hit1.release();
    c->nee.release(), c->prev_accum.release(), c->prev_depth.release();
    c->unit_cost.release(), c->launch_order.release(), c->order_keys.release();
    c->order_sort.keys_out.release(), c->order_sort.vals_a.release(), c->order_sort.vals_b.release(), c->order_sort.counts.release(), c->order_sort.row_total.release();
    c->sort_keys.release(), c->sort_perm.release();
    for (auto& t : c->sort_temp) t.keys_out.release(), t.vals_a.release(), t.vals_b.release(), t.counts.release(), t.row_total.release();
    c->shadow_keys.release(), c->shadow_perm.release();
    c->image_f4.release(), c->state_md.release(), c->state_ray.release(), c->gather.release(), c->sum_accum.release();
}

// Camera::reproject (cuda_camera.cuh:390-426) for one pixel of the frame that has just had its first pass: the first hit point —
// the pixel-centre ray (generateSimpleRay) at the stored depth — seen from the previous camera; where that camera's depth buffer holds
// the same surface (within 1 % of the distance), its accumulator * blend is appended to this pixel's.
struct PrevCamera {
    float position[3], x_axis[3], y_axis[3], z_axis[3];
    float tan_half_fov, aspect_ratio;
};
__global__ void __launch_bounds__(256) rz_reproject_kernel(const DFrame f, const DCamera cam, const PrevCamera prev, const float4* prev_accum,
                                                           const float* prev_depth, float blend) {
    const PixelId p = pixel_of_thread(f, cam, blockIdx.x, threadIdx.x);
    if (!p.active) return;
    Ray ray;
    generate_simple_ray(cam, ray, p.x, p.y);
    const v3 space_p = ray.o + ray.d * f.depth[p.local];
    const v3 rel = space_p - ld3(prev.position);
    const v3 local_p = transform_backward(ld3(prev.x_axis), ld3(prev.y_axis), ld3(prev.z_axis), rel);
    if (local_p.z <= 0.0f) return;  // behind the previous camera
    const float fx = (((local_p.x / local_p.z) / prev.tan_half_fov) + 0.5f) * float(cam.width);
    const float fy = (((local_p.y / local_p.z) / (-prev.tan_half_fov / prev.aspect_ratio)) + 0.5f) * float(cam.height);
    if (fx < 0.0f || fx >= float(cam.width) || fy < 0.0f || fy >= float(cam.height)) return;  // outside the previous frustum
    const uint32_t sx = uint32_t(fx), sy = uint32_t(fy);
    // the history is a row-major image of the WHOLE previous frame (keep_history): the source pixel of a moved camera may have been
    // rendered by another device of the context.  Pixels no shard of this context owns hold depth 0 and never pass the test below.
    const size_t from = size_t(sy) * cam.width + sx;
    const float point_dist = magnitude(rel), buffer_dist = prev_depth[from];
    if (fabsf(point_dist - buffer_dist) < 0.01f * point_dist) {
        const float4 a = f.accum[p.local], h = prev_accum[from];
        f.accum[p.local] = make_float4(a.x + h.x * blend, a.y + h.y * blend, a.z + h.z * blend, a.w + h.w * blend);
    }
}

// Called where a first pass is about to run.  Returns whether the finished first pass is to be followed by reproject_after_first_pass.
// The frame a restart replaces is kept as row-major images of the whole frame (accumulator, first-hit depth).  A multi-device head has
// assembled them from all its devices before the call fanned out (assemble_history, history_ready); a single context untiles its own
// shard — pixels of shards rendered elsewhere (hiprz_set_shard by the caller: another process) stay zero and carry no history.
bool keep_history(hiprz_ctx* c) {
    const bool reproject = (c->mode_flags & HIPRZ_COMPAT_REPROJECTION) && c->frame_started && c->n_local_tiles != 0u;
    if (reproject && !c->history_ready) {
        const size_t n = size_t(c->camera.width) * c->camera.height;
        if (c->prev_accum.resize(n) != hipSuccess || c->prev_depth.resize(n) != hipSuccess) return false;
        if (c->world > 1u) {
            (void)hipMemsetAsync(c->prev_accum.ptr, 0, n * sizeof(float4), c->stream);
            (void)hipMemsetAsync(c->prev_depth.ptr, 0, n * sizeof(float), c->stream);
        }
        RZ_LAUNCH((rz_untile_kernel<float4>), dim3(c->n_local_tiles), dim3(256), 0, c->stream, c->accum.ptr, c->prev_accum.ptr, c->camera.width,
                           c->camera.height, c->tiles_x, c->rank, c->world);
        RZ_LAUNCH((rz_untile_kernel<float>), dim3(c->n_local_tiles), dim3(256), 0, c->stream, c->depth.ptr, c->prev_depth.ptr, c->camera.width,
                           c->camera.height, c->tiles_x, c->rank, c->world);
    }
    c->history_ready = false;
    return reproject;
}
// Multi-device head, before a render call fans out to devices that are about to restart their frames: the whole previous frame —
// every device's tiles over the peer-to-peer path of the readbacks — as row-major images on the head, then a copy to every peer.
int assemble_history(hiprz_ctx* c) {
    if (c->peers.empty() || !(c->mode_flags & HIPRZ_COMPAT_REPROJECTION) || !c->reset_pending || !c->frame_started || !c->have_camera) return HIPRZ_OK;
    if (c->shard_mode == HIPRZ_SHARD_SAMPLES) return HIPRZ_OK;  // every part holds the context's whole share: each keeps its own history (keep_history)
    (void)hipSetDevice(c->device);
    const size_t n = size_t(c->camera.width) * c->camera.height;
    RZ_HIP(c, c->prev_accum.resize(n));
    RZ_HIP(c, c->prev_depth.resize(n));
    if (c->user_world > 1u) {
        RZ_HIP(c, hipMemsetAsync(c->prev_accum.ptr, 0, n * sizeof(float4), c->stream));
        RZ_HIP(c, hipMemsetAsync(c->prev_depth.ptr, 0, n * sizeof(float), c->stream));
    }
    const uint32_t n_parts = uint32_t(c->peers.size()) + 1u;
    const size_t stride = size_t(c->n_local_tiles) * 256u;
    RZ_HIP(c, c->gather.resize(stride * n_parts * (sizeof(float4) + sizeof(float))));
    float4* parts_a = reinterpret_cast<float4*>(c->gather.ptr);
    float* parts_d = reinterpret_cast<float*>(parts_a + stride * n_parts);
    if (c->n_local_tiles) {
        RZ_HIP(c, hipMemcpyAsync(parts_a, c->accum.ptr, stride * sizeof(float4), hipMemcpyDeviceToDevice, c->stream));
        RZ_HIP(c, hipMemcpyAsync(parts_d, c->depth.ptr, stride * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    }
    for (uint32_t r = 1; r < n_parts; ++r) {
        hiprz_ctx* p = c->peers[r - 1u];
        const size_t local = size_t(p->n_local_tiles) * 256u;
        if (!local) continue;
        (void)hipSetDevice(p->device);
        RZ_HIP(c, hipMemcpyPeerAsync(parts_a + stride * r, c->device, p->accum.ptr, p->device, local * sizeof(float4), p->stream));
        RZ_HIP(c, hipMemcpyPeerAsync(parts_d + stride * r, c->device, p->depth.ptr, p->device, local * sizeof(float), p->stream));
        RZ_HIP(c, hipEventRecord(p->peer_done, p->stream));
        (void)hipSetDevice(c->device);
        RZ_HIP(c, hipStreamWaitEvent(c->stream, p->peer_done, 0));
    }
    if (c->n_local_tiles) {
        RZ_LAUNCH((rz_untile_gathered_kernel<float4>), dim3(c->n_local_tiles, n_parts), dim3(256), 0, c->stream, parts_a, stride, c->prev_accum.ptr,
                           c->camera.width, c->camera.height, c->tiles_x, c->tiles_x * c->tiles_y, c->world, c->rank);
        RZ_LAUNCH((rz_untile_gathered_kernel<float>), dim3(c->n_local_tiles, n_parts), dim3(256), 0, c->stream, parts_d, stride, c->prev_depth.ptr,
                           c->camera.width, c->camera.height, c->tiles_x, c->tiles_x * c->tiles_y, c->world, c->rank);
    }
    c->history_ready = true;
    // every peer gets the same images; its stream waits for the copy before its first pass runs
    if (!c->history_done) RZ_HIP(c, hipEventCreateWithFlags(&c->history_done, hipEventDisableTiming));
    for (hiprz_ctx* p : c->peers) {
        (void)hipSetDevice(p->device);
        RZ_HIP(c, p->prev_accum.resize(n));
        RZ_HIP(c, p->prev_depth.resize(n));
        (void)hipSetDevice(c->device);
        RZ_HIP(c, hipMemcpyPeerAsync(p->prev_accum.ptr, p->device, c->prev_accum.ptr, c->device, n * sizeof(float4), c->stream));
        RZ_HIP(c, hipMemcpyPeerAsync(p->prev_depth.ptr, p->device, c->prev_depth.ptr, c->device, n * sizeof(float), c->stream));
        p->history_ready = true;
    }
    RZ_HIP(c, hipEventRecord(c->history_done, c->stream));
    for (hiprz_ctx* p : c->peers) {
        (void)hipSetDevice(p->device);
        RZ_HIP(c, hipStreamWaitEvent(p->stream, c->history_done, 0));
    }
    (void)hipSetDevice(c->device);
    return HIPRZ_OK;
}
void reproject_after_first_pass(hiprz_ctx* c, const DFrame& f, const hiprz_camera& previous) {
    PrevCamera prev;
    std::memcpy(prev.position, previous.position, 12), std::memcpy(prev.x_axis, previous.x_axis, 12);
    std::memcpy(prev.y_axis, previous.y_axis, 12), std::memcpy(prev.z_axis, previous.z_axis, 12);
    prev.tan_half_fov = previous.tan_half_fov, prev.aspect_ratio = previous.aspect_ratio;
    const PassGeometry g = pass_geometry(c);
    RZ_LAUNCH(rz_reproject_kernel, g.grid, dim3(256), 0, c->stream, f, c->dcamera, prev, c->prev_accum.ptr, c->prev_depth.ptr, c->temporal_blend);
}

int allocate_frame(hiprz_ctx* c) {
    c->frame_started = false;  // whatever history there was belongs to other buffers
    const uint32_t W = c->camera.width, H = c->camera.height;
    c->tiles_x = (W + 31u) / 32u;
    c->tiles_y = (H + 7u) / 8u;
    c->n_local_tiles = shard_local_tiles(c->tiles_x, c->tiles_y, c->rank, c->world);
    // owned active pixels (ray counter of this shard)
    uint64_t owned = 0;
    for (uint32_t lt = 0; lt < c->n_local_tiles; ++lt) {
        uint32_t tx, ty;
        shard_tile(lt, c->tiles_x, c->rank, c->world, tx, ty);  // (hiprz_shard.hpp)
        const uint32_t w = std::min(32u, W - tx * 32u), h = std::min(8u, H - ty * 8u);
        owned += uint64_t(w) * h;
    }
    c->owned_pixels = owned;
    const size_t n = size_t(c->n_local_tiles) * 256u;
    RZ_HIP(c, c->st0.resize(n));
    RZ_HIP(c, c->st1.resize(n));
    RZ_HIP(c, c->st2.resize(n));
    RZ_HIP(c, c->accum.resize(n));
    RZ_HIP(c, c->hit0.resize(n));
    RZ_HIP(c, c->hit1.resize(n));
    RZ_HIP(c, c->sort_keys.resize(n));
    RZ_HIP(c, c->sort_perm.resize(n));
    RZ_HIP(c, c->shadow_keys.resize(n));
    RZ_HIP(c, c->shadow_perm.resize(n));
    if (n) {
        RZ_HIP(c, hipMemsetAsync(c->sort_keys.ptr, 0, n * sizeof(uint32_t), c->stream));
        RZ_HIP(c, hipMemsetAsync(c->shadow_keys.ptr, 0, n * sizeof(uint32_t), c->stream));
        const int src = sort_workspace(c, n);
        if (src != HIPRZ_OK) return src;
    }
    RZ_HIP(c, c->depth.resize(n));
    RZ_HIP(c, c->rgba8.resize(n));
    RZ_HIP(c, c->image_f4.resize(size_t(W) * H));
    {   // heaviest-first launch order of the resident kernels: at most one unit per wave of the shard
        const size_t units = size_t(c->n_local_tiles) * 4u;
        RZ_HIP(c, c->unit_cost.resize(units));
        RZ_HIP(c, c->launch_order.resize(units));
        RZ_HIP(c, c->order_keys.resize(units));
        if (units) {
            RZ_HIP(c, hipMemsetAsync(c->unit_cost.ptr, 0, units * sizeof(uint32_t), c->stream));
            const int orc = sort_temp_resize(c, c->order_sort, units);
            if (orc != HIPRZ_OK) return orc;
        }
        c->order_units = 0u, c->batches_since_order = 0u;
    }
    if (n) {
        RZ_HIP(c, hipMemsetAsync(c->accum.ptr, 0, n * sizeof(float4), c->stream));
        RZ_HIP(c, hipMemsetAsync(c->depth.ptr, 0, n * sizeof(float), c->stream));
        RZ_HIP(c, hipMemsetAsync(c->rgba8.ptr, 0, n * sizeof(uint32_t), c->stream));
        RZ_HIP(c, hipMemsetAsync(c->st0.ptr, 0, n * sizeof(float4), c->stream));
        RZ_HIP(c, hipMemsetAsync(c->st1.ptr, 0, n * sizeof(float4), c->stream));
        RZ_HIP(c, hipMemsetAsync(c->st2.ptr, 0, n * sizeof(float2), c->stream));
    }
    return HIPRZ_OK;
}

bool resident_active(const hiprz_ctx* c) { return c->pipeline == 2; }

}  // namespace

namespace hiprz {

// rays are reordered where the walk is bound by scattered fetches: scenes not staged in LDS, split pipeline
bool sort_enabled(const hiprz_ctx* c) {
    if (c->pipeline != 1 || c->sort_rays == 0) return false;
    if (c->sort_rays == 1) return true;
    // measured (1920x1080+, MODE 3; the sort itself costs ~0.12 ms per pass at 1080p): many small instances (config E, 46) 33.9 ->
    // 26.5 ms per pass; one mid-size mesh (config C, 12 k nodes) trace kernel 853 -> 645 us, step 7.70 -> 6.98 ms; one big mesh
    // (config D, 600 k nodes) trace kernel 2 656 -> 2 586 us but step 22.5 -> 23.2 ms.  So: on for many instances, and for trees
    // small enough that a coherent wave finds its nodes in LDS / L2.
    return !use_lds_scene(c) && effective_mode(c) >= 3 && (c->dscene.n_instances >= 16u || c->n_nodes <= kLatencyBoundNodes);
}

DFrame make_frame(hiprz_ctx* c, bool counted) {
    DFrame f{};
    f.st0 = c->st0.ptr, f.st1 = c->st1.ptr, f.st2 = c->st2.ptr;
    f.accum = c->accum.ptr, f.depth = c->depth.ptr, f.rgba8 = c->rgba8.ptr;
    f.hit0 = c->hit0.ptr, f.hit1 = c->hit1.ptr;
    f.pass = c->pass_dev.ptr;
    f.counters = counted ? c->counters_dev.ptr : nullptr;
    f.tiles_x = c->tiles_x, f.rank = c->rank, f.world = c->world, f.n_local_tiles = c->n_local_tiles;
    f.xcd_swizzle = c->xcd_swizzle ? 1u : 0u;
    f.nee = c->nee.ptr, f.nee_quads = 4u + 2u * (c->config.spot_samples + c->config.direct_samples);
    const bool sorting = sort_enabled(c);
    f.sort_key = sorting ? c->sort_keys.ptr : nullptr;
    f.perm = sorting ? c->sort_perm.ptr : nullptr;
    const bool shadow_sorting = sorting && c->shadow_sort != 0 && c->pipeline == 1 && defer_shadows(c);
    f.shadow_key = shadow_sorting ? c->shadow_keys.ptr : nullptr;
    f.shadow_perm = shadow_sorting ? c->shadow_perm.ptr : nullptr;
    // resident kernels: units by falling cost once an order for THIS kind of unit exists; costs are always collected (not while counting)
    const uint32_t units = c->pipeline == 2 ? (wave_resident(c) ? c->n_local_tiles * 4u : c->n_local_tiles) : 0u;
    f.unit_cost = c->heavy_first && units && !counted ? c->unit_cost.ptr : nullptr;
    f.launch_order = c->heavy_first && units && !counted && c->order_units == units ? c->launch_order.ptr : nullptr;
    return f;
}

DConfig make_config(const hiprz_ctx* c) {
    return DConfig{c->config.max_depth, c->config.spot_samples, c->config.direct_samples, c->config.seed, c->mode_flags};
}

// The binned walk pays off when a mesh visit is short and uniform (every mesh tree is a single leaf, e.g. the
// Cornell configs: 329 vs 370 us per pass); with deep mesh trees a round lasts as long as its slowest item
// and the nested walk is faster (config C: 1 668 vs 2 450 us).
int effective_mode(const hiprz_ctx* c) {
    if (c->scene_tree != HIPRZ_TREE_REFERENCE || (c->mode_flags & kIntegratorFlags)) return 3;  // (compat integrator: the cooperative walk, or the fused kernel's skip-link walk)  // rebuilt trees: the front-to-back cooperative walks only
    if (c->traversal_mode >= 0) return c->traversal_mode;
    // records do not fit LDS: skip-link walks in single-wave workgroups
    if (!c->lds_scene && c->pipeline == 1) return 3;
    return c->dscene.mesh_stack_entries <= 2u ? 2 : 1;
}

// Shadow rays get their own kernel when the scene has lights, is not staged in LDS (split pipeline) and the sample slots of a
// segment fit the 30-bit mask of the hand-over record.
bool defer_shadows(const hiprz_ctx* c) {
    return c->defer_shadow_rays && c->pipeline == 1 && !use_lds_scene(c) && c->dscene.n_spot_lights + c->dscene.n_direct_lights != 0u &&
           c->config.spot_samples + c->config.direct_samples <= 30u;
}

// the resident pipeline on a scene that is not staged in LDS: per-wave chains of passes (rz_wave_batch_kernel); needs the front-to-back
// walk (mode 3) and a scene without lights (shadow rays are deferred to a kernel of their own otherwise)
bool wave_resident(const hiprz_ctx* c) {
    return c->pipeline == 2 && !use_lds_scene(c) && c->dscene.n_spot_lights + c->dscene.n_direct_lights == 0u && c->nolight_kernels && c->walk_order != 0 &&
           (c->traversal_mode == -1 || c->traversal_mode == 3);
}

void resolve_pipeline(hiprz_ctx* c) {
    const int before = c->pipeline;
    // a shard small enough to be ONE round of waves on the chip pays the slowest wave of every kernel of every pass in the split
    // pipeline; without lights it runs per-wave chains of passes instead (hiprz_kernels.hpp: rz_wave_batch_kernel)
    const bool dark_capable = c->have_scene && !use_lds_scene(c) && c->dscene.n_spot_lights + c->dscene.n_direct_lights == 0u && c->nolight_kernels && c->walk_order != 0 &&
                              (c->traversal_mode == -1 || c->traversal_mode == 3);
    const bool small_dark_shard = dark_capable && c->have_camera && c->n_local_tiles != 0u && c->n_local_tiles * 4u <= c->wave_resident_max;
    if (c->mode_flags & kIntegratorFlags) c->pipeline = c->pipeline_setting == 0 ? 0 : 1;  // CUDA-compat integrator: split (sorted rays, cooperative walks, deferred shadow rays); 0 = one fused kernel per pass
    else if (c->scene_tree != HIPRZ_TREE_REFERENCE)  // rebuilt trees: the front-to-back cooperative walks only (split, or per-wave resident)
        c->pipeline = dark_capable && (c->pipeline_setting == 2 || (c->pipeline_setting < 0 && small_dark_shard)) ? 2 : 1;
    else if (c->pipeline_setting >= 0) c->pipeline = c->pipeline_setting;
    else if (small_dark_shard) c->pipeline = 2;
    else {
        // resident needs blob + walk workspace + 8 KiB of parked state per workgroup, four workgroups per CU
        const size_t lds = size_t(c->dscene.hot_bytes) + size_t(c->stack_entries) * 1024u + BinnedLds::kFixedBytes + 8u * 1024u;
        const bool mode_ok = c->traversal_mode == -1 || c->traversal_mode == 1 || c->traversal_mode == 2;  // walks the batch kernel has
        c->pipeline = (c->have_scene && c->lds_scene && c->lds_scene_override != 0 && mode_ok && lds <= 40u * 1024u) ? 2 : 1;
    }
    if (c->pipeline != before) invalidate_graphs(c);
}

bool use_lds_scene(const hiprz_ctx* c) {
    if (c->lds_scene_override == 0 || c->scene_tree != HIPRZ_TREE_REFERENCE || (c->mode_flags & kIntegratorFlags)) return false;
    if (c->lds_scene_override == 1) return size_t(c->dscene.hot_bytes) + size_t(c->stack_entries) * 1024u <= 160u * 1024u;
    return c->lds_scene;
}


// Two radix passes (16 key bits) are enough while a bin of the coarser order still holds a wave's worth of rays: up to ~2 M owned
// pixels (config C: step 4.53 -> 4.33 ms, the sort 88 -> 59 us per pass).  Bigger frames and scenes with lights (whose shadow rays
// follow a sorted order of their own) keep all 24 bits (config E: 16 bits 55.9 ms per step against 51.0).
int effective_sort_bits(const hiprz_ctx* c) {
    if (c->sort_bits > 0) return c->sort_bits;
    const bool lights = c->dscene.n_spot_lights + c->dscene.n_direct_lights != 0u;
    return (!lights && size_t(c->n_local_tiles) * 256u <= (size_t(32) << 16)) ? 16 : 24;
}

PassGeometry pass_geometry(const hiprz_ctx* c) {
    PassGeometry g;
    // with the XCD swizzle the grid is padded to a multiple of 8 workgroups (the extra ones find no tile)
    g.grid = dim3(c->xcd_swizzle ? ((c->n_local_tiles + 7u) / 8u) * 8u : c->n_local_tiles), g.block = dim3(256);
    g.lds_scene = use_lds_scene(c);
    g.blob = g.lds_scene ? c->dscene.hot_bytes : 0u;
    g.mode = effective_mode(c);
    if (g.mode >= 3 && (g.lds_scene || (c->pipeline != 1 && !wave_resident(c)))) g.mode = 1;  // skip links are for scenes that are not staged whole: trace kernel, wave batch kernel
    g.stack_lds = size_t(c->stack_entries) * 256u * sizeof(uint32_t);
    g.walk_lds = g.mode == 2 ? size_t(BinnedLds::bytes_host(c->dscene.world_stack_entries, c->dscene.mesh_stack_entries)) : g.mode == 1 ? g.stack_lds : 0u;
    return g;
}

}  // namespace hiprz

namespace {

// one pass on the stream: trace + shade (split pipeline) or the fused kernel
void launch_pass(hiprz_ctx* c, const DFrame& f, bool first, bool counted, hipEvent_t between_trace_and_shade = nullptr) {
    c->sorted_this_pass = false;
    if (c->pipeline == 1 || wave_resident(c)) {  // (the first pass of a wave-resident frame: the split kernels)
        launch_trace(c, f, first, counted);
        if (between_trace_and_shade) (void)hipEventRecord(between_trace_and_shade, c->stream);
        launch_shade(c, f, first, counted);
    } else {
        launch_fused(c, f, first, counted);
    }
}

// resident pipeline: all `n` cumulative passes of the batch in one launch (+ one launch that advances the pass index)
void launch_resident(hiprz_ctx* c, const DFrame& f, uint32_t n, bool counted, hipEvent_t before = nullptr, hipEvent_t after = nullptr) {
    launch_batch(c, f, n, counted, before, after);
    RZ_LAUNCH(rz_pass_add_kernel, dim3(1), dim3(1), 0, c->stream, c->pass_dev.ptr, n);
    c->rgba8_valid = true;
    // Heaviest first: the batch that just ran left every unit's cost behind.  The order is derived after the first batch that measured
    // (and whenever the kind of unit changed) and refreshed every 64th batch — per-tile costs of a fixed view are stable, paths
    // regenerate at the same pixels — by one key kernel + a 24-bit radix sort of a few thousand keys on the render stream.
    if (f.unit_cost && n >= 2u) {
        const uint32_t units = wave_resident(c) ? c->n_local_tiles * 4u : c->n_local_tiles;
        c->batches_since_order += 1u;
        if (c->order_units != units || c->batches_since_order >= 64u) {
            RZ_LAUNCH(rz_order_keys_kernel, dim3((units + 255u) / 256u), dim3(256), 0, c->stream, c->unit_cost.ptr, c->order_keys.ptr, units);
            sort_u32(c->stream, c->order_keys.ptr, units, 24, c->launch_order.ptr, nullptr, c->order_sort);
            c->order_units = units, c->batches_since_order = 0u;
        }
    }
}

hipEvent_t take_event(hiprz_ctx* c) {
    if (!c->event_pool.empty()) {
        hipEvent_t e = c->event_pool.back();
        c->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

void drop_graph(hiprz_ctx* c) {
    // a replay of the old exec may still be in flight on the stream (an asynchronous caller that changes a setting between
    // two render calls): wait for it before the exec goes away
    if (c->graph_exec && c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->graph_exec) (void)hipGraphExecDestroy(c->graph_exec);
    c->graph_exec = nullptr;
    c->graph_valid = false;
}

// [cumulative pass, sort, pass update] x n on the stream — eagerly, or into a capture
void enqueue_cumulative(hiprz_ctx* c, const DFrame& f, uint32_t n) {
    for (uint32_t i = 0; i < n; ++i) {
        launch_pass(c, f, false, false);
        launch_sort(c);
        RZ_LAUNCH(rz_pass_update_kernel, dim3(1), dim3(1), 0, c->stream, c->pass_dev.ptr);
    }
}

int finish_batch(hiprz_ctx* c, hipEvent_t e0, hipEvent_t e1, uint32_t n_passes, const StageTimer& timer) {
    RZ_HIP(c, hipEventRecord(e1, c->stream));
    RZ_HIP(c, hipGetLastError());
    if (c->pending_events.size() >= 4096) {  // nobody is collecting timings: recycle the oldest pair
        c->event_pool.push_back(c->pending_events.front().first);
        c->event_pool.push_back(c->pending_events.front().second);
        c->pending_events.erase(c->pending_events.begin());
        c->pending_launches.erase(c->pending_launches.begin());
    }
    c->pending_events.emplace_back(e0, e1);
    c->pending_launches.push_back(n_passes);
    c->timings.set("render (enqueue)", timer.ms());
    return HIPRZ_OK;
}

// What a captured batch depends on, byte for byte: the arguments every kernel of the batch receives by value (scene, camera, config and
// frame views: raw pointers into buffers the context owns, sizes, sharding), the workspaces the sorts use, and every setting that
// selects a kernel instantiation or a grid size.  The graph holds kernel nodes and event fork / join nodes only — no memset, memcpy or
// library nodes, no host pointers — so "same key" means a replay does what an eager batch would do now.
std::vector<unsigned char> graph_key_of(hiprz_ctx* c, const DFrame& f, uint32_t n_passes) {
    std::vector<unsigned char> key;
    auto put = [&key](const void* p, size_t n) { key.insert(key.end(), static_cast<const unsigned char*>(p), static_cast<const unsigned char*>(p) + n); };
    const DConfig cfg = make_config(c);
    put(&c->dscene, sizeof(DScene)), put(&c->dcamera, sizeof(DCamera)), put(&cfg, sizeof cfg), put(&f, sizeof f);
    for (const auto& t : c->sort_temp) {
        const void* ptrs[5] = {t.keys_out.ptr, t.vals_a.ptr, t.vals_b.ptr, t.counts.ptr, t.row_total.ptr};
        put(ptrs, sizeof ptrs);
    }
    const int settings[] = {c->pipeline, effective_mode(c), c->walk_order, c->trace_waves, int(c->scene_tree), effective_sort_bits(c), int(defer_shadows(c)), int(c->shadow_packet),
                            int(use_lds_scene(c)), int(c->flat_world), c->nolight_kernels, int(c->n_textures), int(c->n_nodes), int(c->xcd_swizzle),
                            int(sort_enabled(c)), c->shadow_sort, c->batch_waves, int(c->stack_entries), int(n_passes), int(c->n_local_tiles)};
    put(settings, sizeof settings);
    return key;
}

int render_impl(hiprz_ctx* c, uint32_t n_passes, bool counted) {
    if (!c->have_scene || !c->have_camera) return fail(c, HIPRZ_ERR_STATE, "render before scene and camera upload");
    if (n_passes == 0 || c->n_local_tiles == 0) return HIPRZ_OK;
    resolve_pipeline(c);  // the choice depends on the selected camera's shard size too
    StageTimer timer;
    if (defer_shadows(c)) {  // hand-over buffers of the deferred shadow rays: (4 + 2 * samples) float4 per owned pixel
        const size_t n = size_t(c->n_local_tiles) * 256u, k = c->config.spot_samples + c->config.direct_samples;
        if (c->nee.count < n * (4u + 2u * k)) c->graph_valid = false;
        RZ_HIP(c, c->nee.resize(n * (4u + 2u * k)));
    }
    const DFrame f = make_frame(c, counted);
    if (!f.perm) c->perm_valid = false;                                        // passes without reordering leave the order behind
    else if (!c->reset_pending && !c->perm_valid) launch_sort_identity(c);    // reordering was switched on between two batches
    hipEvent_t e0 = take_event(c), e1 = take_event(c);
    RZ_HIP(c, hipEventRecord(e0, c->stream));
    c->rgba8_valid = false;
    if (c->pipeline == 2) {
        uint32_t remaining = n_passes;
        if (c->reset_pending) {  // renderFirstPass: the fused kernel
            const bool history = keep_history(c);
            const hiprz_camera previous = c->frame_camera;
            c->frame_camera = c->camera, c->frame_started = true;
            RZ_LAUNCH(rz_pass_reset_kernel, dim3(1), dim3(1), 0, c->stream, c->pass_dev.ptr);
            launch_pass(c, f, true, counted);
            if (history) reproject_after_first_pass(c, f, previous);
            RZ_LAUNCH(rz_pass_update_kernel, dim3(1), dim3(1), 0, c->stream, c->pass_dev.ptr);
            c->reset_pending = false;
            c->passes = 1;
            c->ray_count = c->owned_pixels;
            remaining -= 1u;
        }
        if (remaining) {
            c->kernel_event_passes = 0;
            if (counted) launch_resident(c, f, remaining, true);
            else if (c->time_kernels) {  // bench.py's roofline: the batch kernel's own duration
                while (c->kernel_events.size() < 3u) {
                    hipEvent_t e = nullptr;
                    (void)hipEventCreate(&e);
                    c->kernel_events.push_back(e);
                }
                launch_resident(c, f, remaining, false, c->kernel_events[0], c->kernel_events[1]);
                c->kernel_event_passes = remaining;
            } else launch_resident(c, f, remaining, false);
            c->passes += remaining;
            c->ray_count += uint64_t(remaining) * c->owned_pixels;
        }
        return finish_batch(c, e0, e1, n_passes, timer);
    }
    if (c->use_graph && !c->time_kernels && !counted && !c->reset_pending && n_passes >= 2) {
        // steady state: one graph launch instead of 2 * n_passes kernel launches
        std::vector<unsigned char> key = graph_key_of(c, f, n_passes);
#ifdef RZ_GRAPH_ASSERT  // debug builds: a key that changed under a graph still marked valid is a missed invalidation — say so loudly
        if (c->graph_valid && c->graph_passes == n_passes && key != c->graph_key) {
            std::fprintf(stderr, "hiprz: captured graph marked valid although its launch arguments changed (a buffer was reallocated or a setting changed without invalidate_graphs)\n");
            std::abort();
        }
#endif
        if (!c->graph_valid || c->graph_passes != n_passes || key != c->graph_key) {
            drop_graph(c);
            c->graph_key = std::move(key);
            hipGraph_t graph = nullptr;
            RZ_HIP(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
            enqueue_cumulative(c, f, n_passes);
            RZ_HIP(c, hipStreamEndCapture(c->stream, &graph));
            const hipError_t ie = hipGraphInstantiate(&c->graph_exec, graph, nullptr, nullptr, 0);
            (void)hipGraphDestroy(graph);
            if (ie != hipSuccess) return fail(c, HIPRZ_ERR_DEVICE, std::string("hipGraphInstantiate: ") + hipGetErrorString(ie));
            c->graph_passes = n_passes;
            c->graph_valid = true;
            c->graph_captures += 1;
        }
        RZ_HIP(c, hipGraphLaunch(c->graph_exec, c->stream));
        c->passes += n_passes;
        c->ray_count += uint64_t(n_passes) * c->owned_pixels;
        return finish_batch(c, e0, e1, n_passes, timer);
    }
    // kernel-level timing (bench.py's roofline): events before the trace kernel, between the two kernels and after the
    // shade kernel of every pass.  Event timing does not work from inside a captured graph, so a timed batch is launched
    // eagerly.
    const bool timed = c->time_kernels && c->pipeline == 1 && !counted && !c->reset_pending;
    if (timed) {
        while (c->kernel_events.size() < size_t(3 * n_passes)) {
            hipEvent_t e = nullptr;
            (void)hipEventCreate(&e);
            c->kernel_events.push_back(e);
        }
        c->kernel_event_passes = n_passes;
    }
    for (uint32_t i = 0; i < n_passes; ++i) {
        if (timed) {
            (void)hipEventRecord(c->kernel_events[3 * i], c->stream);
            launch_pass(c, f, false, false, c->kernel_events[3 * i + 1]);
            (void)hipEventRecord(c->kernel_events[3 * i + 2], c->stream);
            launch_sort(c);
            RZ_LAUNCH(rz_pass_update_kernel, dim3(1), dim3(1), 0, c->stream, c->pass_dev.ptr);
            c->passes += 1;
            c->ray_count += c->owned_pixels;
            continue;
        }
        if (c->reset_pending) {
            const bool history = keep_history(c);
            const hiprz_camera previous = c->frame_camera;
            c->frame_camera = c->camera, c->frame_started = true;
            RZ_LAUNCH(rz_pass_reset_kernel, dim3(1), dim3(1), 0, c->stream, c->pass_dev.ptr);
            launch_pass(c, f, true, counted);
            if (history) reproject_after_first_pass(c, f, previous);
            c->reset_pending = false;
            c->passes = 0;
            c->ray_count = 0;
        } else {
            launch_pass(c, f, false, counted);
        }
        launch_sort(c);
        RZ_LAUNCH(rz_pass_update_kernel, dim3(1), dim3(1), 0, c->stream, c->pass_dev.ptr);
        c->passes += 1;
        c->ray_count += c->owned_pixels;  // traced_rays += W*H per pass (cpu_engine_renderer.cpp:173), per shard
    }
    return finish_batch(c, e0, e1, n_passes, timer);
}

template <typename T, typename PeerTiles>
int read_untiled(hiprz_ctx* c, const T* tiles, T* dst, size_t bytes, const char* what, PeerTiles peer_tiles_of) {
    if (!c->have_camera) return fail(c, HIPRZ_ERR_STATE, "readback before camera upload");
    const size_t n = size_t(c->camera.width) * c->camera.height;
    if (!dst || bytes != n * sizeof(T)) return fail(c, HIPRZ_ERR_INVALID, std::string(what) + ": destination size mismatch");
    StageTimer timer;
    T* image = reinterpret_cast<T*>(c->image_f4.ptr);
    // the shards of this context cover the whole frame unless the caller split it further (hiprz_set_shard): only then are there
    // pixels nobody writes, and only then is the image cleared first
    if (c->user_world > 1u) RZ_HIP(c, hipMemsetAsync(image, 0, bytes, c->stream));
    if (c->peers.empty() || c->shard_mode == HIPRZ_SHARD_SAMPLES) {  // (sample mode: `tiles` is the head's own / the summed buffer of the whole share)
        if (c->n_local_tiles)
            RZ_LAUNCH((rz_untile_kernel<T>), dim3(c->n_local_tiles), dim3(256), 0, c->stream, tiles, image,
                               c->camera.width, c->camera.height, c->tiles_x, c->rank, c->world);
    } else {
        // multi-device head: every device's tiles land in a slice of their own of ONE staging buffer — each peer pushes its slice on
        // ITS stream, behind its own rendering, so the copies of different peers cross their xGMI links side by side — and one launch
        // on the head's stream, which waits for all of them, untiles every slice (shard rank0 + r of `world` in slice r)
        const uint32_t n_parts = uint32_t(c->peers.size()) + 1u;
        const size_t stride = size_t(c->n_local_tiles) * 256u;  // the head owns the lowest rank of the context: no shard has more tiles
        RZ_HIP(c, c->gather.resize(stride * n_parts * sizeof(T)));
        T* parts = reinterpret_cast<T*>(c->gather.ptr);
        if (c->n_local_tiles) RZ_HIP(c, hipMemcpyAsync(parts, tiles, stride * sizeof(T), hipMemcpyDeviceToDevice, c->stream));
        for (uint32_t r = 1; r < n_parts; ++r) {
            hiprz_ctx* p = c->peers[r - 1u];
            const size_t peer_bytes = size_t(p->n_local_tiles) * 256u * sizeof(T);
            if (!peer_bytes) continue;
            (void)hipSetDevice(p->device);
            RZ_HIP(c, hipMemcpyPeerAsync(parts + stride * r, c->device, peer_tiles_of(p), p->device, peer_bytes, p->stream));
            RZ_HIP(c, hipEventRecord(p->peer_done, p->stream));
            (void)hipSetDevice(c->device);
            RZ_HIP(c, hipStreamWaitEvent(c->stream, p->peer_done, 0));
        }
        if (c->n_local_tiles)
            RZ_LAUNCH((rz_untile_gathered_kernel<T>), dim3(c->n_local_tiles, n_parts), dim3(256), 0, c->stream, parts, stride, image, c->camera.width,
                               c->camera.height, c->tiles_x, c->tiles_x * c->tiles_y, c->world, c->rank);
    }
    RZ_HIP(c, hipMemcpyAsync(dst, image, bytes, hipMemcpyDeviceToHost, c->stream));
    RZ_HIP(c, hipStreamSynchronize(c->stream));
    c->timings.set(what, timer.ms());
    return HIPRZ_OK;
}

// Everything the kernels will dereference is checked here, on the host, before any launch: a
// bad index or a cyclic tree would otherwise fault or hang the GPU.  Also derives the skip links,
// the world-tree leaves and the tree depths the upload needs.
struct SceneCheck {
    std::string error;
    std::vector<uint8_t> reachable;  // nodes some walk can get to (a snapshot may hold others: they are never followed)
    std::vector<uint32_t> skip;
    std::vector<uint32_t> world_leaves;
    uint32_t world_depth = 0, mesh_depth = 0;
};
int check_scene(const hiprz_scene* sc, SceneCheck& out) {
    auto bad = [&out](const std::string& m) {
        out.error = m;
        return HIPRZ_ERR_INVALID;
    };
    if (!sc) return bad("scene is null");
    // ---- validate everything the kernels will dereference, on the host, before any launch ----
    if (sc->n_materials < 2 || !sc->materials) return bad("scene needs materials[0]=world, [1]=default");
    if (sc->n_materials > 65536u) return bad("more than 65536 materials");
    if ((sc->n_nodes && !sc->nodes) || (sc->n_tris && (!sc->tris || !sc->tri_attrs)) || (sc->n_instances && !sc->instances) ||
        (sc->n_tlas_order && !sc->tlas_order) || (sc->n_inst_materials && !sc->inst_materials) ||
        (sc->n_textures && !sc->textures) || (sc->texel_bytes && !sc->texels) || (sc->n_spot_lights && !sc->spot_lights) ||
        (sc->n_direct_lights && !sc->direct_lights))
        return bad("upload_scene: null array with non-zero count");
    for (uint32_t i = 0; i < sc->n_textures; ++i) {
        const hiprz_texture& t = sc->textures[i];
        const uint64_t texel = t.kind == HIPRZ_TEX_R8 ? 1u : 4u;
        if (t.kind > HIPRZ_TEX_R32F || t.width == 0 || t.height == 0 || (t.offset & 3u) ||
            uint64_t(t.offset) + texel * t.width * t.height > sc->texel_bytes)
            return bad("texture " + std::to_string(i) + ": bad kind/size/offset");
    }
    auto tex_ok = [&](int32_t t, uint32_t kind) { return t < 0 || (uint32_t(t) < sc->n_textures && sc->textures[t].kind == kind); };
    for (uint32_t i = 0; i < sc->n_materials; ++i) {
        const hiprz_material& m = sc->materials[i];
        if (!tex_ok(m.texture, HIPRZ_TEX_RGBA8) || !tex_ok(m.normal_map, HIPRZ_TEX_RGBA8) ||
            !tex_ok(m.metalness_map, HIPRZ_TEX_R8) || !tex_ok(m.roughness_map, HIPRZ_TEX_R8) ||
            !tex_ok(m.emission_map, HIPRZ_TEX_R32F))
            return bad("material " + std::to_string(i) + ": map index/kind invalid");
    }
    for (uint32_t i = 0; i < sc->n_inst_materials; ++i)
        if (sc->inst_materials[i] >= int32_t(sc->n_materials))
            return bad("inst_materials[" + std::to_string(i) + "] out of range");
    for (uint32_t i = 0; i < sc->n_tlas_order; ++i)
        if (sc->tlas_order[i] >= sc->n_instances) return bad("tlas_order entry out of range");
    for (uint32_t i = 0; i < sc->n_instances; ++i) {
        const hiprz_instance& in = sc->instances[i];
        if (in.material_count > 64u || uint64_t(in.material_base) + in.material_count > sc->n_inst_materials)
            return bad("instance " + std::to_string(i) + ": material table out of range");
    }
    std::vector<uint32_t> skip(sc->n_nodes ? sc->n_nodes : 1, RZ_END);
    TreeCheck check{sc, skip, std::vector<uint8_t>(sc->n_nodes ? sc->n_nodes : 1, 0)};
    uint32_t world_depth = 0, mesh_depth = 0;
    if (sc->n_instances) {
        if (!check.walk(sc->tlas_root, true)) return bad("world tree: " + check.error);
        world_depth = check.max_depth;
        std::vector<uint8_t> root_seen(sc->n_nodes, 0);
        for (uint32_t i = 0; i < sc->n_tlas_order; ++i) {
            const uint32_t root = sc->instances[sc->tlas_order[i]].blas_root;
            if (root >= sc->n_nodes) return bad("instance mesh root out of range");
            if (root_seen[root]) continue;
            root_seen[root] = 1;
            check.max_depth = 0;
            if (!check.walk(root, false)) return bad("mesh tree: " + check.error);
            mesh_depth = std::max(mesh_depth, check.max_depth);
        }
    }
    out.skip = std::move(skip);
    out.reachable = std::move(check.visited);
    out.world_leaves = std::move(check.world_leaves);
    out.world_depth = world_depth, out.mesh_depth = mesh_depth;
    return HIPRZ_OK;
}

// Device-side tables derived from a validated scene (pure host): relayouted nodes + links (reference order and per octant).
struct DerivedTables {
    std::vector<uint32_t> new_index;
    std::vector<hiprz_node> dnodes;
    std::vector<uint32_t> dskip;
    std::vector<uint32_t> dskip8;  // [node][octant]: skip links of the front-to-back mesh walk (hiprz_device.hpp: fetch_node_ordered)
};
int derive_tables(const hiprz_scene* sc, SceneCheck& chk, DerivedTables& out) {
    // Relayout: breadth-first over ALL trees at once (world root, then every distinct mesh root, then their child
    // pairs, ...), children staying adjacent.  The levels nearest the roots become a prefix of the array (the part
    // MODE 3 caches in LDS) and siblings/cousins share cache lines.  Leaf ranges are untouched.
    std::vector<uint32_t>& new_index = out.new_index;
    new_index.assign(sc->n_nodes, RZ_END);
    std::vector<uint32_t> bfs;
    bfs.reserve(sc->n_nodes);
    auto enqueue = [&](uint32_t old) {
        if (old < sc->n_nodes && new_index[old] == RZ_END) {
            new_index[old] = uint32_t(bfs.size());
            bfs.push_back(old);
        }
    };
    if (sc->n_instances) enqueue(sc->tlas_root);
    for (uint32_t i = 0; i < sc->n_tlas_order; ++i) enqueue(sc->instances[sc->tlas_order[i]].blas_root);
    // Child pairs follow the roots.  A 64-byte record pair is one 128-byte cache line when it starts at an even index: one empty slot
    // behind an odd number of roots puts every pair on a line of its own, so that the second child — visited after the first one's
    // subtree, or probed together with it — is on the line the first one brought in.
    const uint32_t pad_at = (bfs.size() & 1u) ? uint32_t(bfs.size()) : RZ_END;
    if (pad_at != RZ_END) bfs.push_back(RZ_END);
    for (size_t q = 0; q < bfs.size(); ++q) {
        if (bfs[q] == RZ_END) continue;
        const hiprz_node& n = sc->nodes[bfs[q]];
        if (!(n.meta & HIPRZ_NODE_LEAF)) enqueue(n.begin), enqueue(n.begin + 1);
    }
    for (uint32_t old = 0; old < sc->n_nodes; ++old) enqueue(old);  // nodes no instance reaches keep a slot
    const size_t n_total = bfs.size();  // the scene's nodes + the padding slot
    std::vector<hiprz_node>& dnodes = out.dnodes;
    std::vector<uint32_t>& dskip = out.dskip;
    hiprz_node empty{};
    empty.meta = HIPRZ_NODE_LEAF;  // no triangles, reached by nothing
    dnodes.assign(n_total ? n_total : 0, empty);
    dskip.assign(n_total ? n_total : 1, RZ_END);
    for (uint32_t old = 0; old < sc->n_nodes; ++old) {
        hiprz_node n = sc->nodes[old];
        if (!(n.meta & HIPRZ_NODE_LEAF)) n.begin = new_index[n.begin];
        dnodes[new_index[old]] = n;
        dskip[new_index[old]] = chk.skip[old] == RZ_END ? RZ_END : new_index[chk.skip[old]];
    }

    // ---- skip links per ray octant (front-to-back walk) ----
    // Under octant o an inner node with partition type p (X=2, Y=1, Z=0) is left towards its SECOND child first when bit p of o is
    // set; a size split (type 3) is never flipped.  The child visited first links to its sibling, the other one inherits the
    // parent's link.  Parents precede their children in the breadth-first numbering, so one ascending sweep fills all tables;
    // roots end their walks (RZ_END).  Octant 0 reproduces dskip.
    std::vector<uint32_t>& dskip8 = out.dskip8;
    dskip8.assign((n_total ? n_total : 1) * 8u, RZ_END);
    for (uint32_t n = 0; n < n_total; ++n) {
        const hiprz_node& nd = dnodes[n];
        if (nd.meta & HIPRZ_NODE_LEAF) continue;
        const uint32_t ptype = (nd.meta >> HIPRZ_NODE_PTYPE_SHIFT) & 3u, c0 = nd.begin;
        if (c0 <= n || size_t(c0) + 1 >= n_total) continue;  // cannot happen after check_scene + the BFS relayout; keeps the sweep safe
        for (uint32_t o = 0; o < 8u; ++o) {
            const uint32_t flip = (o >> ptype) & 1u;  // ptype 3 reads bit 3 = 0
            dskip8[size_t(c0 + flip) * 8u + o] = c0 + 1u - flip;
            dskip8[size_t(c0 + 1u - flip) * 8u + o] = dskip8[size_t(n) * 8u + o];
        }
    }

    // The kernels follow these derived tables blindly: prove on the host that every walk over them terminates
    // (each step moves strictly forward in depth-first order, so a walk may take at most one step per node).
    {
        auto terminates = [](const std::vector<hiprz_node>& nodes, const std::vector<uint32_t>& links, uint32_t root) {
            uint32_t n = root;
            for (size_t steps = 0; steps <= nodes.size(); ++steps) {
                if (n == RZ_END) return true;
                if (n >= nodes.size()) return false;
                const hiprz_node& nd = nodes[n];
                n = !(nd.meta & HIPRZ_NODE_LEAF) ? nd.begin : links[n];
            }
            return false;
        };
        bool ok = true;
        for (uint32_t old = 0; ok && old < sc->n_nodes; ++old)  // the stack walks reach the second child as first + 1
            if (!(sc->nodes[old].meta & HIPRZ_NODE_LEAF)) ok = new_index[sc->nodes[old].begin + 1] == new_index[sc->nodes[old].begin] + 1u;
        if (ok && sc->n_instances) ok = terminates(dnodes, dskip, new_index[sc->tlas_root]);
        for (uint32_t i = 0; ok && i < sc->n_tlas_order; ++i) {
            const uint32_t root = new_index[sc->instances[sc->tlas_order[i]].blas_root];
            ok = terminates(dnodes, dskip, root);
        }
        // the same for every octant's links: a walk that enters every box takes exactly one step per node of the tree it walks
        auto terminates8 = [&](uint32_t root, uint32_t o) {
            uint32_t n = root;
            for (size_t steps = 0; steps <= dnodes.size(); ++steps) {
                if (n == RZ_END) return true;
                if (n >= dnodes.size()) return false;
                const hiprz_node& nd = dnodes[n];
                if (!(nd.meta & HIPRZ_NODE_LEAF)) n = nd.begin + ((o >> ((nd.meta >> HIPRZ_NODE_PTYPE_SHIFT) & 3u)) & 1u);
                else n = dskip8[size_t(n) * 8u + o];
            }
            return false;
        };
        for (uint32_t old = 0; ok && old < sc->n_nodes; ++old)  // octant 0 is the reference's order (nodes no walk reaches have no links to compare)
            if (old < chk.reachable.size() && chk.reachable[old]) ok = dskip8[size_t(new_index[old]) * 8u] == dskip[new_index[old]];
        if (ok && sc->n_instances) ok = terminates8(new_index[sc->tlas_root], 0u);
        {
            std::vector<uint8_t> seen(n_total ? n_total : 1, 0);
            for (uint32_t i = 0; ok && i < sc->n_tlas_order; ++i) {
                const uint32_t root = new_index[sc->instances[sc->tlas_order[i]].blas_root];
                if (seen[root]) continue;
                seen[root] = 1;
                for (uint32_t o = 0; ok && o < 8u; ++o) ok = terminates8(root, o);
            }
        }
        if (!ok) {
            chk.error = "internal: derived walk tables are inconsistent (refusing to launch)";
            return HIPRZ_ERR_INVALID;
        }
    }
    return HIPRZ_OK;
}

}  // namespace

namespace {
// hiprz_select_camera on one context: the selected camera's frame state lives in the context itself, the others are parked
void select_camera_one(hiprz_ctx* c, uint32_t k) {
    if (k == c->active_camera || k >= c->parked.size()) return;
    hiprz_frame_state& self = *c;
    std::swap(c->parked[c->active_camera], self);  // park the active one (its slot held an empty state)
    std::swap(self, c->parked[k]);
    c->active_camera = k;
}
// A captured graph of a batch of passes stays valid while nothing its kernel arguments depend on has changed: the setters
// invalidate it only when a value really differs (both host sides call hiprz_set_config before every frame).
template <typename T>
void assign_setting(hiprz_ctx* c, T& field, const T& value) {
    if (std::memcmp(&field, &value, sizeof(T)) != 0) {
        field = value;
        invalidate_graphs(c);
    }
}
}  // namespace

// Tile-major hand-off of a context's share.  One device / one stream: the owned tiles of shard (rank, world), in order.  A context over
// several devices or streams (n parts) hands out n slices of equal capacity — slice r holds sub-shard rank * n + r of world * n, what
// that device rendered — so the slices of all the ranks of a job, laid end to end, are the sub-shards 0 .. world * n - 1 in order:
// hiprz_untile_gathered with world * n parts of that capacity assembles the frame.
namespace {
size_t part_capacity(const hiprz_ctx* c) {  // pixels per slice: the largest sub-shard of the job (the lowest ranks own one tile more)
    return size_t(shard_local_tiles(c->tiles_x, c->tiles_y, 0u, c->world)) * 256u;  // c->world is already user_world * parts on a multi-device head
}
bool samples_head(const hiprz_ctx* c) { return c->shard_mode == HIPRZ_SHARD_SAMPLES && !c->peers.empty(); }
// HIPRZ_SHARD_SAMPLES head: out = the sum of the parts' accumulators over the context's share (tile-major, n_local_tiles * 256 pixels).
// Every peer pushes its accumulators into its slice of the head's staging buffer on ITS stream, behind its own rendering (peer-to-peer
// over xGMI for another device), the head's stream waits for all of them and one launch adds them up in part order.  A peer's next push
// waits for that launch (sum_done): nothing here synchronises with the host.
int sum_parts(hiprz_ctx* c, float4* out) {
    const size_t n = size_t(c->n_local_tiles) * 256u;
    if (!n) return HIPRZ_OK;
    (void)hipSetDevice(c->device);
    const uint32_t n_staged = uint32_t(c->peers.size());
    RZ_HIP(c, c->gather.resize(n * n_staged * sizeof(float4)));
    float4* staged = reinterpret_cast<float4*>(c->gather.ptr);
    if (!c->sum_done) RZ_HIP(c, hipEventCreateWithFlags(&c->sum_done, hipEventDisableTiming));
    for (uint32_t r = 0; r < n_staged; ++r) {
        hiprz_ctx* p = c->peers[r];
        if (p->n_local_tiles != c->n_local_tiles || !p->accum.ptr) return fail(c, HIPRZ_ERR_STATE, "sample sharding: a part's share differs from the head's");
        (void)hipSetDevice(p->device);
        if (c->sum_recorded) RZ_HIP(c, hipStreamWaitEvent(p->stream, c->sum_done, 0));
        RZ_HIP(c, hipMemcpyPeerAsync(staged + n * r, c->device, p->accum.ptr, p->device, n * sizeof(float4), p->stream));
        RZ_HIP(c, hipEventRecord(p->peer_done, p->stream));
        (void)hipSetDevice(c->device);
        RZ_HIP(c, hipStreamWaitEvent(c->stream, p->peer_done, 0));
    }
    RZ_LAUNCH(rz_sum_parts_kernel, dim3(c->n_local_tiles), dim3(256), 0, c->stream, c->accum.ptr, staged, n, n_staged, out, uint32_t(n));
    RZ_HIP(c, hipGetLastError());
    RZ_HIP(c, hipEventRecord(c->sum_done, c->stream));
    c->sum_recorded = true;
    return HIPRZ_OK;
}
template <typename T, typename Tiles>
int export_tiles(hiprz_ctx* c, void* dst_device, size_t bytes, const char* what, Tiles tiles_of) {
    if (c->shard_mode == HIPRZ_SHARD_SAMPLES) {  // one slice: the head's buffer of the whole share (the caller summed / tone-mapped the parts into it)
        const size_t own = size_t(c->n_local_tiles) * 256u * sizeof(T);
        if (!dst_device || bytes < own) return fail(c, HIPRZ_ERR_INVALID, std::string(what) + ": destination too small");
        (void)hipSetDevice(c->device);
        if (own) RZ_HIP(c, hipMemcpyAsync(dst_device, tiles_of(c), own, hipMemcpyDeviceToDevice, c->stream));
        return HIPRZ_OK;
    }
    const uint32_t n_parts = uint32_t(c->peers.size()) + 1u;
    const size_t cap = c->peers.empty() ? size_t(c->n_local_tiles) * 256u : part_capacity(c);
    if (!dst_device || bytes < cap * n_parts * sizeof(T)) return fail(c, HIPRZ_ERR_INVALID, std::string(what) + ": destination too small");
    (void)hipSetDevice(c->device);
    T* dst = static_cast<T*>(dst_device);
    if (c->n_local_tiles) RZ_HIP(c, hipMemcpyAsync(dst, tiles_of(c), size_t(c->n_local_tiles) * 256u * sizeof(T), hipMemcpyDeviceToDevice, c->stream));
    for (uint32_t r = 1; r < n_parts; ++r) {  // every peer pushes its slice on its own stream; the head's stream waits for all of them
        hiprz_ctx* p = c->peers[r - 1u];
        const size_t peer_bytes = size_t(p->n_local_tiles) * 256u * sizeof(T);
        if (!peer_bytes) continue;
        (void)hipSetDevice(p->device);
        RZ_HIP(c, hipMemcpyPeerAsync(dst + cap * r, c->device, tiles_of(p), p->device, peer_bytes, p->stream));
        RZ_HIP(c, hipEventRecord(p->peer_done, p->stream));
        (void)hipSetDevice(c->device);
        RZ_HIP(c, hipStreamWaitEvent(c->stream, p->peer_done, 0));
    }
    return HIPRZ_OK;
}
}  // namespace

extern "C" {

int hiprz_validate_scene(const hiprz_scene* scene, char* message, size_t len) {
    SceneCheck chk;
    int rc = check_scene(scene, chk);
    if (rc == HIPRZ_OK) {  // also prove that the tables the kernels will follow can be derived and terminate
        DerivedTables derived;
        rc = derive_tables(scene, chk, derived);
    }
    if (message && len) std::snprintf(message, len, "%s", chk.error.c_str());
    return rc;
}

int hiprz_create(hiprz_ctx** out, int device_id) {
    if (!out) return fail(nullptr, HIPRZ_ERR_INVALID, "hiprz_create: out is null");
    *out = nullptr;
    int n_devices = 0;
    hipError_t e = hipGetDeviceCount(&n_devices);
    if (e != hipSuccess || n_devices == 0)
        return fail(nullptr, HIPRZ_ERR_DEVICE, std::string("no HIP device: ") + hipGetErrorString(e));
    if (device_id < 0 || device_id >= n_devices) return fail(nullptr, HIPRZ_ERR_INVALID, "hiprz_create: device id out of range");
    e = hipSetDevice(device_id);
    if (e != hipSuccess) return fail(nullptr, HIPRZ_ERR_DEVICE, std::string("hipSetDevice: ") + hipGetErrorString(e));
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device_id);
    if (e != hipSuccess) return fail(nullptr, HIPRZ_ERR_DEVICE, std::string("hipGetDeviceProperties: ") + hipGetErrorString(e));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, HIPRZ_ERR_DEVICE, std::string("hiprz is built for gfx950 only, device is ") + prop.gcnArchName);
    if (const int rc = resolve_kernels(device_id); rc != HIPRZ_OK) return rc;
    auto* c = new hiprz_ctx();
    if (const char* w = std::getenv("HIPRZ_TRACE_WAVES")) c->trace_waves = std::atoi(w);
    if (const char* w = std::getenv("HIPRZ_DEFER_SHADOWS")) c->defer_shadow_rays = std::atoi(w) != 0;
    if (const char* w = std::getenv("HIPRZ_HEAVY_FIRST")) c->heavy_first = std::atoi(w) != 0;
    if (const char* w = std::getenv("HIPRZ_BATCH_WAVES")) c->batch_waves = std::atoi(w);
    if (const char* w = std::getenv("HIPRZ_NOLIGHT_KERNELS")) c->nolight_kernels = std::atoi(w) != 0;
    if (const char* w = std::getenv("HIPRZ_SORT_BITS")) c->sort_bits = std::min(24, std::max(0, std::atoi(w)));
    if (const char* w = std::getenv("HIPRZ_SHADOW_PACKET")) c->shadow_packet = std::atoi(w);
    if (const char* w = std::getenv("HIPRZ_SHADOW_TREE")) c->shadow_tree = std::atoi(w) != 0;
    if (const char* w = std::getenv("HIPRZ_SHADOW_SORT")) c->shadow_sort = std::atoi(w) != 0;
    if (const char* w = std::getenv("HIPRZ_WAVE_RESIDENT_MAX")) c->wave_resident_max = uint32_t(std::max(0, std::atoi(w)));
    c->device = device_id;
    c->parked.resize(1);  // one camera; its state lives in the context itself
    e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->peer_done, hipEventDisableTiming);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->aux_fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->aux_join, hipEventDisableTiming);
    if (e == hipSuccess) e = c->pass_dev.resize(1);
    if (e == hipSuccess) e = c->counters_dev.resize(16);
    if (e == hipSuccess) e = c->pick_dev.resize(4);
    if (e == hipSuccess) e = hipMemsetAsync(c->pass_dev.ptr, 0, sizeof(uint32_t), c->stream);
    if (e != hipSuccess) {
        const std::string msg = std::string("context setup: ") + hipGetErrorString(e);
        hiprz_destroy(c);
        return fail(nullptr, HIPRZ_ERR_DEVICE, msg);
    }
    *out = c;
    return HIPRZ_OK;
}

int hiprz_create_multi(hiprz_ctx** out, const int* device_ids, int n_devices) {
    if (!out) return fail(nullptr, HIPRZ_ERR_INVALID, "hiprz_create_multi: out is null");
    *out = nullptr;
    if (!device_ids || n_devices < 1 || n_devices > 64) return fail(nullptr, HIPRZ_ERR_INVALID, "hiprz_create_multi: 1..64 device ids");
    hiprz_ctx* head = nullptr;
    int rc = hiprz_create(&head, device_ids[0]);
    if (rc != HIPRZ_OK) return rc;
    for (int r = 1; r < n_devices; ++r) {
        hiprz_ctx* peer = nullptr;
        rc = hiprz_create(&peer, device_ids[r]);
        if (rc != HIPRZ_OK) {
            const std::string msg = g_create_error;
            (void)hiprz_destroy(head);
            return fail(nullptr, rc, msg);
        }
        head->peers.push_back(peer);
        if (device_ids[r] != device_ids[0]) {  // direct copies between the two GPUs (xGMI); absent peer access hip stages them through the host
            int can = 0;
            (void)hipDeviceCanAccessPeer(&can, device_ids[0], device_ids[r]);
            if (can) {
                (void)hipSetDevice(device_ids[0]);
                (void)hipDeviceEnablePeerAccess(device_ids[r], 0);
                (void)hipGetLastError();  // "already enabled" is fine
            }
        }
    }
    rc = hiprz_set_shard(head, 0u, 1u);
    if (rc != HIPRZ_OK) {
        const std::string msg = head->error;
        (void)hiprz_destroy(head);
        return fail(nullptr, rc, msg);
    }
    *out = head;
    return HIPRZ_OK;
}

int hiprz_device_count(hiprz_ctx* c, uint32_t* out) {
    if (!c || !out) return HIPRZ_ERR_INVALID;
    *out = uint32_t(c->peers.size()) + 1u;
    return HIPRZ_OK;
}

// ---- cameras: the reference renders every enabled camera of the world per call (cpu_engine_renderer.cpp:97-117) ----
int hiprz_set_camera_count(hiprz_ctx* c, uint32_t n) {
    if (!c) return HIPRZ_ERR_INVALID;
    RZ_FANOUT(c, hiprz_set_camera_count(p, n));
    if (n == 0u || n > 4096u) return fail(c, HIPRZ_ERR_INVALID, "set_camera_count: 1..4096 cameras");
    (void)hipSetDevice(c->device);
    if (c->active_camera >= n) select_camera_one(c, 0u);
    RZ_HIP(c, hipStreamSynchronize(c->stream));
    for (uint32_t k = n; k < uint32_t(c->parked.size()); ++k) {
        if (c->parked[k].graph_exec) (void)hipGraphExecDestroy(c->parked[k].graph_exec);
        release_frame(&c->parked[k]);
        c->parked[k].pass_dev.release();
    }
    c->parked.resize(n);
    return HIPRZ_OK;
}

int hiprz_camera_count(hiprz_ctx* c, uint32_t* out) {
    if (!c || !out) return HIPRZ_ERR_INVALID;
    *out = uint32_t(c->parked.size());
    return HIPRZ_OK;
}

int hiprz_select_camera(hiprz_ctx* c, uint32_t index) {
    if (!c) return HIPRZ_ERR_INVALID;
    RZ_FANOUT(c, hiprz_select_camera(p, index));
    if (index >= c->parked.size()) return fail(c, HIPRZ_ERR_INVALID, "select_camera: index beyond hiprz_set_camera_count");
    select_camera_one(c, index);
    if (!c->pass_dev.ptr) {  // a camera selected for the first time: its device-resident pass index
        (void)hipSetDevice(c->device);
        RZ_HIP(c, c->pass_dev.resize(1));
        RZ_HIP(c, hipMemsetAsync(c->pass_dev.ptr, 0, sizeof(uint32_t), c->stream));
    }
    return HIPRZ_OK;
}

int hiprz_destroy(hiprz_ctx* c) {
    if (!c) return HIPRZ_OK;
    for (hiprz_ctx* p : c->peers) (void)hiprz_destroy(p);
    c->peers.clear();
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (uint32_t k = 0; k < uint32_t(c->parked.size()); ++k) {  // every camera's frame
        if (k == c->active_camera) continue;
        if (c->parked[k].graph_exec) (void)hipGraphExecDestroy(c->parked[k].graph_exec);
        release_frame(&c->parked[k]);
        c->parked[k].pass_dev.release();
    }
    if (c->peer_done) (void)hipEventDestroy(c->peer_done);
    if (c->history_done) (void)hipEventDestroy(c->history_done);
    if (c->sum_done) (void)hipEventDestroy(c->sum_done);
    if (c->aux_stream) (void)hipStreamSynchronize(c->aux_stream), (void)hipStreamDestroy(c->aux_stream);
    if (c->aux_fork) (void)hipEventDestroy(c->aux_fork);
    if (c->aux_join) (void)hipEventDestroy(c->aux_join);
    drop_graph(c);
    for (auto& p : c->pending_events) {
        (void)hipEventDestroy(p.first);
        (void)hipEventDestroy(p.second);
    }
    for (auto e : c->event_pool) (void)hipEventDestroy(e);
    for (auto e : c->kernel_events) (void)hipEventDestroy(e);
    c->hot.release(), c->node_skip.release(), c->nodes64.release(), c->textures.release();
    c->dev_nodes.release(), c->has_mesh.release(), c->build_temp.release(), c->slot_parent.release(), c->ref_to_dev.release(), c->refit_visit.release();
    c->world_items.release(), c->update_tris.release(), c->update_attrs.release();
    c->shadow_nodes64.release(), c->shadow_order.release();
    c->build_sort.keys_out.release(), c->build_sort.vals_a.release(), c->build_sort.vals_b.release(), c->build_sort.counts.release(), c->build_sort.row_total.release();
    c->texels.release(), c->spot_lights.release(), c->direct_lights.release();
    release_frame(c);
    c->pass_dev.release(), c->counters_dev.release(), c->pick_dev.release();
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return HIPRZ_OK;
}

const char* hiprz_last_error(const hiprz_ctx* c) { return c ? c->error.c_str() : g_create_error.c_str(); }

namespace {
int build_shadow_world_tree(hiprz_ctx* c, const std::vector<hiprz_instance>& dinst, DScene& d);
}

int hiprz_upload_scene(hiprz_ctx* c, const hiprz_scene* sc) {
    if (!c) return HIPRZ_ERR_INVALID;
    RZ_FANOUT_OTHER_DEVICES(c, hiprz_upload_scene(p, sc));
    for (hiprz_ctx* p : c->peers)  // streams on this device may still be reading the buffers this call replaces
        if (p->device == c->device) (void)hipStreamSynchronize(p->stream);
    struct Share {  // whatever way this call ends, the streams on this device see what the head holds then
        hiprz_ctx* c;
        ~Share() {
            share_scene_with_streams(c);
            for (hiprz_ctx* p : c->peers)
                if (p->device == c->device) resolve_pipeline(p);
        }
    } share{c};
    invalidate_graphs(c);
    StageTimer timer;
    SceneCheck chk;
    if (check_scene(sc, chk) != HIPRZ_OK) return fail(c, HIPRZ_ERR_INVALID, "upload_scene: " + chk.error);
    // opt-in mesh trees of better quality (hiprz_set_tree): the snapshot is rewritten — new nodes, triangles and their attributes in
    // the new leaf order, every triangle remembering its position in the reference's order — and then takes the usual way
    hiprz_scene rebuilt_scene;
    std::vector<hiprz_node> rebuilt_nodes;
    std::vector<hiprz_tri> rebuilt_tris;
    std::vector<hiprz_tri_attr> rebuilt_attrs;
    std::vector<hiprz_instance> rebuilt_instances;
    uint32_t tree = c->tree_mode;
    c->build_sah = c->device_sah;
    if (tree == HIPRZ_TREE_AUTO) {
        // a scene whose records can be staged in LDS keeps the snapshot's trees (the resident kernels walk those); any other gets the
        // device's surface-area trees.  (A lower bound of the hot blob: node records, triangles + shading records, instances.)
        const size_t records = size_t(sc->n_nodes) * 32u + size_t(sc->n_tris) * 144u + size_t(sc->n_instances) * 112u;
        tree = records > kLdsSceneLimit ? HIPRZ_TREE_DEVICE : HIPRZ_TREE_REFERENCE;
        c->build_sah = true;
    }
    std::vector<uint32_t> order, roots;
    uint32_t n_nodes = 0u, tlas_root = 0u;
    bool identity_order = false;
    if (tree != HIPRZ_TREE_REFERENCE && sc->n_tris != 0u) {
        const uint32_t max_nodes = sc->n_nodes + 2u * sc->n_tris + sc->n_instances + 1u;
        rebuilt_nodes.resize(max_nodes);
        order.resize(sc->n_tris), roots.resize(sc->n_instances ? sc->n_instances : 1u);
        if (hiprz_rebuild_mesh_trees(sc, tree, rebuilt_nodes.data(), max_nodes, &n_nodes, order.data(), roots.data(), &tlas_root) != HIPRZ_OK) {
            if (c->tree_mode != HIPRZ_TREE_AUTO)
                return fail(c, HIPRZ_ERR_INVALID, "upload_scene: the mesh trees could not be rebuilt (leaves of a mesh must tile one range of triangles)");
            tree = HIPRZ_TREE_REFERENCE;  // HIPRZ_TREE_AUTO promises an upload wherever the snapshot's own trees are valid
        }
    }
    if (tree != HIPRZ_TREE_REFERENCE && sc->n_tris != 0u) {
        rebuilt_nodes.resize(n_nodes);
        // (the placeholder trees of a device build keep the snapshot's order when its meshes lie in first-use order, as the hosts'
        // flatteners lay them out: then the 144 bytes per triangle are not copied, and position i is what triangle i is ranked by)
        identity_order = true;
        for (uint32_t i = 0; i < sc->n_tris && identity_order; ++i) identity_order = order[i] == i;
        if (!identity_order) {
            rebuilt_tris.resize(sc->n_tris), rebuilt_attrs.resize(sc->n_tris);
            for (uint32_t i = 0; i < sc->n_tris; ++i) {
                rebuilt_tris[i] = sc->tris[order[i]];
                rebuilt_tris[i].pad0 = order[i];
                rebuilt_attrs[i] = sc->tri_attrs[order[i]];
            }
        }
        rebuilt_instances.assign(sc->instances, sc->instances + sc->n_instances);
        for (uint32_t i = 0; i < sc->n_instances; ++i) rebuilt_instances[i].blas_root = roots[i];
        rebuilt_scene = *sc;
        rebuilt_scene.n_nodes = n_nodes, rebuilt_scene.nodes = rebuilt_nodes.data(), rebuilt_scene.tlas_root = tlas_root;
        if (!identity_order) rebuilt_scene.tris = rebuilt_tris.data(), rebuilt_scene.tri_attrs = rebuilt_attrs.data();
        rebuilt_scene.instances = rebuilt_instances.data();
        sc = &rebuilt_scene;
        if (check_scene(sc, chk) != HIPRZ_OK) return fail(c, HIPRZ_ERR_INVALID, "upload_scene: rebuilt trees: " + chk.error);
        c->timings.set("rebuild mesh trees", timer.ms());
    }
    const bool own_trees = sc == &rebuilt_scene;
    const uint32_t world_depth = chk.world_depth, mesh_depth = chk.mesh_depth;

    DerivedTables derived;
    if (derive_tables(sc, chk, derived) != HIPRZ_OK) return fail(c, HIPRZ_ERR_INVALID, "upload_scene: " + chk.error);
    // from here on the device buffers of the previous scene are being replaced: until the new one is complete there is no scene
    // (a failed upload must not leave the old scene's kernels arguments pointing at reallocated buffers)
    c->have_scene = false;
    c->stack_entries = world_depth + mesh_depth + 2u;
    c->dscene.world_stack_entries = world_depth + 1u;
    c->dscene.mesh_stack_entries = mesh_depth + 1u;
    std::vector<uint32_t>& new_index = derived.new_index;
    std::vector<hiprz_node>& dnodes = derived.dnodes;
    std::vector<uint32_t>& dskip = derived.dskip;
    // shared-reciprocal division is exact only for coordinates that are 0 or in [2^-60, 2^40)
    auto coord_ok = [](float x) {
        uint32_t b;
        std::memcpy(&b, &x, 4);
        const uint32_t e = (b >> 23) & 0xFFu;
        return (b & 0x7FFFFFFFu) == 0u || (e >= 127u - 60u && e < 127u + 40u);
    };
    bool fast_div = true;
    for (const auto& n : dnodes)
        for (int a = 0; a < 3; ++a) fast_div = fast_div && coord_ok(n.bb_min[a]) && coord_ok(n.bb_max[a]);
    for (uint32_t i = 0; i < sc->n_instances; ++i)
        for (int a = 0; a < 3; ++a) fast_div = fast_div && coord_ok(sc->instances[i].bb_min[a]) && coord_ok(sc->instances[i].bb_max[a]);

    (void)hipSetDevice(c->device);
    // hot blob: one buffer, 16-B aligned sections
    std::vector<uint8_t> blob;
    auto append = [&blob](const void* src, size_t bytes) {
        const uint32_t off = uint32_t(blob.size());
        blob.resize(blob.size() + ((bytes + 15u) & ~size_t(15)), 0);
        if (bytes) std::memcpy(blob.data() + off, src, bytes);
        return off;
    };
    DScene& d = c->dscene;
    // device copies keep every box interleaved, (min.x, max.x, min.y, max.y, min.z, max.z), so one axis' two
    // plane distances are one packed operand of the box test (hiprz_device.hpp: box_hit)
    auto interleave = [](float* mn, float* mx) {
        const float v[6] = {mn[0], mx[0], mn[1], mx[1], mn[2], mx[2]};
        mn[0] = v[0], mn[1] = v[1], mn[2] = v[2], mx[0] = v[3], mx[1] = v[4], mx[2] = v[5];
    };
    for (auto& n : dnodes) interleave(n.bb_min, n.bb_max);  // bb_min[3] and bb_max[3] are contiguous
    std::vector<hiprz_instance> dinstances(sc->instances, sc->instances + sc->n_instances);
    for (auto& in : dinstances) {
        if (in.blas_root < sc->n_nodes) in.blas_root = new_index[in.blas_root];
        in.pad0 = (in.scale[0] == 1.0f && in.scale[1] == 1.0f && in.scale[2] == 1.0f) ? 1u : 0u;  // x / 1.0f == x: the walk skips it
        const float v[6] = {in.bb_min[0], in.bb_max[0], in.bb_min[1], in.bb_max[1], in.bb_min[2], in.bb_max[2]};
        in.bb_min[0] = v[0], in.bb_min[1] = v[1], in.bb_min[2] = v[2];
        std::memcpy(&in.pad2, &v[3], 4);
        in.bb_max[0] = v[4], in.bb_max[1] = v[5], in.bb_max[2] = 0.0f;
    }
    d.off_nodes = append(dnodes.data(), sizeof(hiprz_node) * dnodes.size());
    d.off_tlas_order = append(sc->tlas_order, sizeof(uint32_t) * sc->n_tlas_order);
    d.off_instances = append(dinstances.data(), sizeof(hiprz_instance) * dinstances.size());
    // device triangles hold v1 and the edges v2 - v1, v3 - v1; v2 and v3 themselves (normal mapping only) move into
    // the padding words of the attribute record
    std::vector<hiprz_tri> dtris(sc->tris, sc->tris + sc->n_tris);
    std::vector<hiprz_tri_attr> dattrs(sc->tri_attrs, sc->tri_attrs + sc->n_tris);
    for (uint32_t i = 0; i < sc->n_tris; ++i) {
        hiprz_tri& t = dtris[i];
        hiprz_tri_attr& a = dattrs[i];
        if (!own_trees || identity_order) t.pad0 = i;  // position in the reference's leaf order: what equally distant hits are ranked by
        a.pad0 = t.v2[0], a.pad1 = t.v2[1], a.pad2 = t.v2[2], a.pad3 = t.v3[0], a.pad4[0] = t.v3[1], a.pad4[1] = t.v3[2];
        for (int k = 0; k < 3; ++k) {
            const float v2 = t.v2[k], v3 = t.v3[k];
            t.v2[k] = v2 - t.v1[k];
            t.v3[k] = v3 - t.v1[k];
        }
    }
    d.off_tris = append(dtris.data(), sizeof(hiprz_tri) * dtris.size());
    d.off_tri_attrs = append(dattrs.data(), sizeof(hiprz_tri_attr) * dattrs.size());
    d.off_materials = append(sc->materials, sizeof(hiprz_material) * sc->n_materials);
    d.off_inst_materials = append(sc->inst_materials, sizeof(int32_t) * sc->n_inst_materials);
    if (blob.size() > 0xFFFFFFF0ull) return fail(c, HIPRZ_ERR_INVALID, "scene geometry exceeds 4 GiB");
    d.hot_bytes = uint32_t(blob.size());

    // HIPRZ_TREE_DEVICE: the node arrays get room behind the uploaded prefix for the world tree (2 * instances + 1 slots) and for every
    // mesh tree (2 * triangles - 1 slots) the device is going to build; regions start at odd slots, their child pairs at even ones
    const bool device_trees = own_trees && tree == HIPRZ_TREE_DEVICE;
    std::vector<DeviceMesh> device_meshes;
    std::vector<uint32_t> instance_mesh(sc->n_instances, RZ_END);
    uint32_t node_capacity = uint32_t(dnodes.size()), world_region = 0u;
    if (device_trees) {
        std::vector<uint32_t> mesh_of_root(sc->n_nodes, RZ_END);
        for (uint32_t i = 0; i < sc->n_instances; ++i) {
            const uint32_t root = sc->instances[i].blas_root;
            if (root >= sc->n_nodes) continue;
            if (mesh_of_root[root] == RZ_END) {
                const hiprz_node& leaf = sc->nodes[root];  // the placeholder of hiprz_rebuild_mesh_trees(.., HIPRZ_TREE_DEVICE, ..): one leaf per mesh
                DeviceMesh m;
                m.tri_first = leaf.begin, m.n_tris = leaf.meta & HIPRZ_NODE_COUNT_MASK;
                m.ref_first = m.n_tris ? (identity_order ? leaf.begin : sc->tris[leaf.begin].pad0) : 0u;
                m.leaf_slot = new_index[root];
                std::memcpy(m.bb_min, leaf.bb_min, 12), std::memcpy(m.bb_max, leaf.bb_max, 12);
                mesh_of_root[root] = uint32_t(device_meshes.size());
                device_meshes.push_back(m);
            }
            instance_mesh[i] = mesh_of_root[root];
        }
        uint32_t cursor = uint32_t(dnodes.size());
        if (!(cursor & 1u)) cursor += 1u;
        world_region = cursor;
        cursor += 2u * sc->n_instances + 1u;
        node_capacity = device_build_regions(device_meshes, cursor);
        dskip.resize(node_capacity, RZ_END);
    }
    RZ_HIP(c, c->hot.assign(blob.data(), blob.size(), c->stream));
    RZ_HIP(c, c->node_skip.assign(dskip.data(), dskip.size(), c->stream));
    if (device_trees) {
        hiprz_node unused{};
        unused.meta = HIPRZ_NODE_LEAF;  // slots no build fills stay empty leaves nothing links to
        std::vector<hiprz_node> all(node_capacity, unused);
        std::copy(dnodes.begin(), dnodes.end(), all.begin());
        RZ_HIP(c, c->dev_nodes.assign(reinterpret_cast<const uint8_t*>(all.data()), all.size() * sizeof(hiprz_node), c->stream));
        RZ_HIP(c, hipStreamSynchronize(c->stream));
    }
    // front-to-back walk: 64-B records, node (interleaved box) + the 8 octant links
    std::vector<uint32_t> nodes64(size_t(node_capacity ? node_capacity : 1) * 16u, RZ_END);
    for (size_t n = 0; n < dnodes.size(); ++n) {
        std::memcpy(&nodes64[n * 16u], &dnodes[n], sizeof(hiprz_node));
        std::memcpy(&nodes64[n * 16u + 8u], &derived.dskip8[n * 8u], 32);
    }
    RZ_HIP(c, c->nodes64.assign(nodes64.data(), nodes64.size(), c->stream));
    RZ_HIP(c, c->textures.assign(sc->textures, sc->n_textures, c->stream));
    RZ_HIP(c, c->texels.assign(sc->texels, sc->texel_bytes, c->stream));
    RZ_HIP(c, c->spot_lights.assign(sc->spot_lights, sc->n_spot_lights, c->stream));
    RZ_HIP(c, c->direct_lights.assign(sc->direct_lights, sc->n_direct_lights, c->stream));
    RZ_HIP(c, hipStreamSynchronize(c->stream));  // host staging vectors and the caller's arrays may go away after return

    d.hot = reinterpret_cast<const float4*>(c->hot.ptr);
    d.nodes = reinterpret_cast<const float4*>(c->hot.ptr + d.off_nodes);
    d.tlas_order = reinterpret_cast<const uint32_t*>(c->hot.ptr + d.off_tlas_order);
    d.instances = reinterpret_cast<const float4*>(c->hot.ptr + d.off_instances);
    d.tris = reinterpret_cast<const float4*>(c->hot.ptr + d.off_tris);
    d.tri_attrs = reinterpret_cast<const float4*>(c->hot.ptr + d.off_tri_attrs);
    d.materials = reinterpret_cast<const float4*>(c->hot.ptr + d.off_materials);
    d.inst_materials = reinterpret_cast<const int32_t*>(c->hot.ptr + d.off_inst_materials);
    d.fast_div = fast_div ? 1u : 0u;
    d.textures = reinterpret_cast<const float4*>(c->textures.ptr);
    d.texels = c->texels.ptr;
    d.spot_lights = reinterpret_cast<const float4*>(c->spot_lights.ptr);
    d.direct_lights = reinterpret_cast<const float4*>(c->direct_lights.ptr);
    d.n_instances = sc->n_instances;
    d.tlas_root = sc->n_instances ? new_index[sc->tlas_root] : 0u;
    d.node_skip = c->node_skip.ptr;
    for (int a = 0; a < 3; ++a) {
        const float lo = sc->n_instances ? sc->nodes[sc->tlas_root].bb_min[a] : 0.0f, hi = sc->n_instances ? sc->nodes[sc->tlas_root].bb_max[a] : 0.0f;
        d.bounds_min[a] = lo;
        d.bounds_scale[a] = hi > lo ? 32.0f / (hi - lo) : 0.0f;
    }
    d.top_count = std::min<uint32_t>(uint32_t(dnodes.size()), kTopCacheNodes);
    d.nodes64 = reinterpret_cast<const float4*>(c->nodes64.ptr);
    c->n_nodes = sc->n_nodes;
    c->flat_world = sc->n_instances != 0u && (sc->nodes[sc->tlas_root].meta & HIPRZ_NODE_LEAF) && (sc->nodes[sc->tlas_root].meta & HIPRZ_NODE_COUNT_MASK) <= 8u;
    c->n_textures = sc->n_textures;
    // mesh walk rounds of at most 4 node steps and 8 triangles per lane (measured: D 3 163 -> 2 891 us, C 935 -> 892 us)
    d.walk_k = 4u, d.walk_l = 8u;
    d.walk_h = 65u;  // never: ending the node phase early for a full triangle step measured no gain (D 1 010 vs 999 us)
    // ray reordering key: where only the closest-hit walk follows the sorted order, the origin's cell interleaved with where the ray is
    // going groups best (config C, the 6-D Morton code of cell and direction: trace kernel 583 -> 503 us; round 4: the direction on the
    // octahedron, E 2 682 -> 2 556 us, C 324 -> 316; then the cell where the ray leaves the world box instead of a direction, E -> 2 400,
    // C -> 299); where the deferred shadow rays follow that order too (HIPRZ_SHADOW_SORT=0) they fan out from the origin cell, so the
    // origin leads (config E: 86.5 ms per step against 92.6)
    d.sort_variant = (sc->n_spot_lights + sc->n_direct_lights) && c->shadow_sort == 0 ? 0u : 4u;
    // the shadow rays' key: the pixel's set of sample slots (+ 0x100: the shadow kernel's loop over the slots is wave-uniform, a slot few of a
    // wave's pixels hold costs the wave a whole walk; E 40.33 -> 39.85 ms), then the light the ray goes to and the origin's cell in a 64^3
    // grid (+ 0x400; E 38.5 -> 37.8 ms against layout 0 — cell, then direction —, which was the best of the layouts: the rays fan out from the cell)
    d.shadow_variant = 0x500u;
    if (const char* v = std::getenv("HIPRZ_SHADOW_KEY")) d.shadow_variant = uint32_t(std::atoi(v));
    if (const char* v = std::getenv("HIPRZ_SORT_KEY")) d.sort_variant = uint32_t(std::atoi(v));
    // The world and instance levels of the cooperative walks ("while-while" one and two levels above the mesh walk; the order in which a
    // lane meets its instances stays the reference's).  World level: a lane steps through up to 1 + 8 nodes of the world tree per round
    // until it HOLDS a leaf with instances, so that the expensive part — the ray into an instance's space, the mesh walk — runs for many
    // lanes at once instead of for the few that happened to reach a leaf in this step (E, 46 instances: trace kernel 3 010 -> 2 678 us,
    // shade + shadow 2 711 -> 2 636; 1 / 2 / 4 / 8 / 64 further steps: 2 992 / 2 892 / 2 798 / 2 712 / 2 716 us).  Instance level: a lane
    // that misses an instance's box tests the next one in the same round (D 836 -> 803 us; with 4 and more D's lanes reach the big mesh
    // in different rounds, each as long as its longest walk: 1 045 us and worse).  profiles/r03/ab_instance_advance.txt, ab_world_advance.txt
    d.walk_advance = 1u;
    d.world_advance = 8u;
    if (const char* v = std::getenv("HIPRZ_WALK_ADVANCE")) d.walk_advance = uint32_t(std::atoi(v));
    if (const char* v = std::getenv("HIPRZ_WORLD_ADVANCE")) d.world_advance = uint32_t(std::atoi(v));
    if (const char* v = std::getenv("HIPRZ_WALK_K")) d.walk_k = uint32_t(std::atoi(v));
    if (const char* v = std::getenv("HIPRZ_WALK_L")) d.walk_l = uint32_t(std::atoi(v));
    if (const char* v = std::getenv("HIPRZ_WALK_H")) d.walk_h = uint32_t(std::atoi(v));
    d.n_spot_lights = sc->n_spot_lights;
    d.n_direct_lights = sc->n_direct_lights;
    // Stage the blob in LDS when three workgroups per CU (the kernel's register-limited residency)
    // still fit into the CU's 160 KiB together with their traversal stacks.
    c->lds_scene = size_t(d.hot_bytes) + size_t(c->stack_entries) * 1024u + BinnedLds::kFixedBytes <= kLdsSceneLimit;
    c->scene_tree = own_trees ? tree : HIPRZ_TREE_REFERENCE;
    if (own_trees) c->lds_scene = false;  // rebuilt trees are walked front to back on skip links only (ties by reference position)
    c->n_tris = sc->n_tris, c->n_tlas_order = sc->n_tlas_order;
    c->device_meshes.clear(), c->instance_mesh.clear();
    if (device_trees) {
        const bool validate = !std::getenv("HIPRZ_TRUST_DEVICE_TREES");
        d.nodes = reinterpret_cast<const float4*>(c->dev_nodes.ptr);
        c->node_capacity = node_capacity, c->world_region = world_region;
        c->device_instances = dinstances;
        std::vector<uint8_t> has_mesh(sc->n_instances ? sc->n_instances : 1u, 0);
        for (uint32_t i = 0; i < sc->n_instances; ++i) has_mesh[i] = instance_mesh[i] != RZ_END ? 1 : 0;
        RZ_HIP(c, c->has_mesh.assign(has_mesh.data(), has_mesh.size(), c->stream));
        RZ_HIP(c, hipStreamSynchronize(c->stream));
        RZ_HIP(c, c->slot_parent.resize(node_capacity));
        RZ_HIP(c, hipMemsetAsync(c->slot_parent.ptr, 0xFF, size_t(node_capacity) * sizeof(uint32_t), c->stream));  // RZ_END: the single leaves of small meshes have no parent
        int rc = HIPRZ_OK;
        if (sc->n_tlas_order) {
            rc = device_build_world_tree(c, validate);
            if (rc != HIPRZ_OK) return rc;
            d.tlas_root = world_region;
        }
        rc = device_build_mesh_trees(c, device_meshes, instance_mesh, validate);
        if (rc != HIPRZ_OK) return rc;
        c->instance_mesh = instance_mesh;
        for (uint32_t i = 0; i < sc->n_instances; ++i)
            if (instance_mesh[i] != RZ_END && c->device_meshes[instance_mesh[i]].region != RZ_END) c->device_instances[i].blas_root = c->device_meshes[instance_mesh[i]].region;
        uint32_t emitted = c->world_slots;
        for (const auto& m : c->device_meshes) emitted += m.n_slots;
        c->n_nodes = emitted;
    }
    {   // the shadow rays' own world tree over the instances of the world tree (those with a mesh)
        std::vector<uint8_t> member(sc->n_instances ? sc->n_instances : 1u, 0);
        for (uint32_t k = 0; k < sc->n_tlas_order; ++k)
            if (sc->tlas_order[k] < sc->n_instances) member[sc->tlas_order[k]] = 1;
        c->world_members.clear();
        for (uint32_t i = 0; i < sc->n_instances; ++i)
            if (member[i]) c->world_members.push_back(i);
        const int src = build_shadow_world_tree(c, dinstances, d);
        if (src != HIPRZ_OK) return src;
    }
    c->have_scene = true;
    resolve_pipeline(c);
    c->reset_pending = true;  // world changed => accumulation restarts (cpu_engine_renderer.cpp:108-112), for every camera
    for (auto& f : c->parked) f.reset_pending = true;
    c->timings.set("upload scene", timer.ms());
    return HIPRZ_OK;
}

// Materials and lights of the uploaded scene changed, geometry did not (the reference's dirty flags per container, updatable.cpp:23-51;
// Cuda::World re-mirrors only modified containers, cuda_world.cu:28-57): the records are replaced in place — no tree is rebuilt,
// re-derived or re-validated.  The material count must be that of the uploaded scene (instances refer to materials by index).
int hiprz_update_shading(hiprz_ctx* c, const hiprz_material* materials, uint32_t n_materials, const hiprz_spot_light* spot_lights,
                         uint32_t n_spot_lights, const hiprz_direct_light* direct_lights, uint32_t n_direct_lights) {
    if (!c) return HIPRZ_ERR_INVALID;
    RZ_FANOUT_OTHER_DEVICES(c, hiprz_update_shading(p, materials, n_materials, spot_lights, n_spot_lights, direct_lights, n_direct_lights));
    struct Share {
        hiprz_ctx* c;
        ~Share() { share_scene_with_streams(c); }
    } share{c};
    for (hiprz_ctx* p : c->peers)  // streams on this device read the records that are about to be replaced
        if (p->device == c->device) (void)hipStreamSynchronize(p->stream);
    if (!c->have_scene) return fail(c, HIPRZ_ERR_STATE, "update_shading before upload_scene");
    const uint32_t uploaded = (c->dscene.off_inst_materials - c->dscene.off_materials) / uint32_t(sizeof(hiprz_material));
    if (!materials || n_materials < 2u || ((n_materials * sizeof(hiprz_material) + 15u) & ~size_t(15)) != size_t(c->dscene.off_inst_materials - c->dscene.off_materials))
        return fail(c, HIPRZ_ERR_INVALID, "update_shading: the scene was uploaded with " + std::to_string(uploaded) + " material slots");
    if ((n_spot_lights && !spot_lights) || (n_direct_lights && !direct_lights)) return fail(c, HIPRZ_ERR_INVALID, "update_shading: null array with non-zero count");
    std::vector<hiprz_texture> tex(c->n_textures);
    (void)hipSetDevice(c->device);
    if (c->n_textures) RZ_HIP(c, hipMemcpy(tex.data(), c->textures.ptr, sizeof(hiprz_texture) * c->n_textures, hipMemcpyDeviceToHost));
    auto tex_ok = [&](int32_t t, uint32_t kind) { return t < 0 || (uint32_t(t) < c->n_textures && tex[t].kind == kind); };
    for (uint32_t i = 0; i < n_materials; ++i) {
        const hiprz_material& m = materials[i];
        if (!tex_ok(m.texture, HIPRZ_TEX_RGBA8) || !tex_ok(m.normal_map, HIPRZ_TEX_RGBA8) || !tex_ok(m.metalness_map, HIPRZ_TEX_R8) ||
            !tex_ok(m.roughness_map, HIPRZ_TEX_R8) || !tex_ok(m.emission_map, HIPRZ_TEX_R32F))
            return fail(c, HIPRZ_ERR_INVALID, "update_shading: material " + std::to_string(i) + ": map index/kind invalid");
    }
    RZ_HIP(c, hipStreamSynchronize(c->stream));
    RZ_HIP(c, hipMemcpy(c->hot.ptr + c->dscene.off_materials, materials, sizeof(hiprz_material) * n_materials, hipMemcpyHostToDevice));
    RZ_HIP(c, c->spot_lights.assign(spot_lights, n_spot_lights, c->stream));
    RZ_HIP(c, c->direct_lights.assign(direct_lights, n_direct_lights, c->stream));
    RZ_HIP(c, hipStreamSynchronize(c->stream));
    c->dscene.spot_lights = reinterpret_cast<const float4*>(c->spot_lights.ptr);
    c->dscene.direct_lights = reinterpret_cast<const float4*>(c->direct_lights.ptr);
    c->dscene.n_spot_lights = n_spot_lights, c->dscene.n_direct_lights = n_direct_lights;
    const bool no_shadow_sort = (n_spot_lights + n_direct_lights) && c->shadow_sort == 0;
    if (!std::getenv("HIPRZ_SORT_KEY")) c->dscene.sort_variant = no_shadow_sort ? 0u : 4u;
    invalidate_graphs(c);
    c->reset_pending = true;  // the world changed: accumulation restarts (cpu_engine_renderer.cpp:108-112), for every camera
    for (auto& f : c->parked) f.reset_pending = true;
    return HIPRZ_OK;
}

// ---- geometry changes without a host-side tree build (scenes uploaded under HIPRZ_TREE_DEVICE; hiprz_build.hip) ----
namespace {
// An in-place change of the scene failed half way (a device-built tree the host refused to prove terminating, a device error): the node
// tables, the triangle order or the instance roots may be part old, part new — there is no scene any more.  hiprz_render then returns
// HIPRZ_ERR_STATE instead of walking tables nobody proved, and the streams that share this device's copy learn the same (Share's destructor).
int scene_lost(hiprz_ctx* c, int rc) {
    c->have_scene = false;
    c->device_meshes.clear(), c->instance_mesh.clear();
    invalidate_graphs(c);
    return rc;
}
// The shadow rays' own world tree.  anyIntersection's answer does not depend on the order in which a ray meets the instances, so the
// wave-level shadow walk (any_hit_packet) need not follow the reference's world tree — built for another purpose: leaves of several
// instances, met in one fixed sequence — and takes a binned-free surface-area tree over the instances' world boxes instead: binary, one
// instance per leaf, the leaf's box being the instance's own (interleaved) box bit for bit, so that the leaf's test is the instance's
// test.  Built on the host from the instance records the device holds (`dinst`: boxes interleaved like nodes), at every upload and every
// hiprz_update_instances: n log^2 n for n instances, 64-byte walk records with the octant-0 skip links the wave-level walk follows,
// root in record 0.  Buffers are sized once per scene (2 n records), so the DScene a captured graph holds stays valid across updates.
int build_shadow_world_tree(hiprz_ctx* c, const std::vector<hiprz_instance>& dinst, DScene& d) {
    d.shadow_nodes64 = nullptr, d.shadow_order = nullptr, d.shadow_root = RZ_END;
    const std::vector<uint32_t>& members = c->world_members;
    // Where it pays (the living room's pass, wave-level walk on this tree / on the reference's / cooperative walk, ms): 40 instances at 4K 4.20 / 4.34 /
    // 4.75, 100 at 4K 5.91 / 6.13 / 6.65, 100 at 1080p 2.09 / 2.17 / 2.28, 300 at 4K 9.00 / 8.84 / 9.29 — a deep binary tree is a long chain of
    // dependent steps for a wave that crosses many instances; beyond 160 the walk keeps the reference's tree (and the host is spared the build).
    if (!c->shadow_tree || members.empty() || members.size() > 160u) return HIPRZ_OK;
    struct Box {
        float mn[3], mx[3];
    };
    auto box_of = [&](uint32_t i) {
        const hiprz_instance& in = dinst[i];
        float max_y;
        std::memcpy(&max_y, &in.pad2, 4);
        return Box{{in.bb_min[0], in.bb_min[2], in.bb_max[0]}, {in.bb_min[1], max_y, in.bb_max[1]}};
    };
    auto grow = [](Box& b, const Box& o) {
        for (int a = 0; a < 3; ++a) b.mn[a] = std::min(b.mn[a], o.mn[a]), b.mx[a] = std::max(b.mx[a], o.mx[a]);
    };
    auto area = [](const Box& b) {
        const float x = b.mx[0] - b.mn[0], y = b.mx[1] - b.mn[1], z = b.mx[2] - b.mn[2];
        return x * y + y * z + z * x;
    };
    const uint32_t n = uint32_t(members.size());
    std::vector<uint32_t> order(members), rec(size_t(2u * n) * 16u, RZ_END), sorted, best;
    std::vector<float> left_area;
    struct Task {
        uint32_t node, lo, hi, link;
    };
    std::vector<Task> stack{{0u, 0u, n, RZ_END}};
    uint32_t next_free = 1u;
    while (!stack.empty()) {
        const Task t = stack.back();
        stack.pop_back();
        Box b = box_of(order[t.lo]);
        for (uint32_t k = t.lo + 1u; k < t.hi; ++k) grow(b, box_of(order[k]));
        const float interleaved[6] = {b.mn[0], b.mx[0], b.mn[1], b.mx[1], b.mn[2], b.mx[2]};
        uint32_t* r = &rec[size_t(t.node) * 16u];
        std::memcpy(r, interleaved, 24);
        for (int o = 0; o < 8; ++o) r[8 + o] = t.link;  // (only the wave-level walk follows this tree: the order of octant 0 under every octant)
        const uint32_t len = t.hi - t.lo;
        if (len == 1u) {
            r[6] = t.lo, r[7] = HIPRZ_NODE_LEAF | 1u;
            continue;
        }
        // the cheapest cut of the instances sorted by box centre along one of the axes: area(left) * |left| + area(right) * |right|
        float best_cost = 3.0e38f;
        uint32_t best_axis = 0u, best_cut = len / 2u;
        for (uint32_t axis = 0; axis < 3u; ++axis) {
            sorted.assign(order.begin() + t.lo, order.begin() + t.hi);
            std::stable_sort(sorted.begin(), sorted.end(), [&](uint32_t x, uint32_t y) {
                const Box bx = box_of(x), by = box_of(y);
                return bx.mn[axis] + bx.mx[axis] < by.mn[axis] + by.mx[axis];
            });
            left_area.assign(len, 0.0f);
            Box acc = box_of(sorted[0]);
            for (uint32_t k = 1u; k < len; ++k) left_area[k] = area(acc), grow(acc, box_of(sorted[k]));  // area of the first k
            acc = box_of(sorted[len - 1u]);
            for (uint32_t k = len - 1u; k >= 1u; --k) {  // cut before position k
                const float cost = left_area[k] * float(k) + area(acc) * float(len - k);
                if (cost < best_cost) best_cost = cost, best_axis = axis, best_cut = k, best = sorted;
                grow(acc, box_of(sorted[k - 1u]));
            }
        }
        if (best.size() != len) best.assign(order.begin() + t.lo, order.begin() + t.hi);
        std::copy(best.begin(), best.end(), order.begin() + t.lo);
        best.clear();
        const uint32_t first = next_free;
        next_free += 2u;
        r[6] = first, r[7] = (2u - best_axis) << HIPRZ_NODE_PTYPE_SHIFT;  // the lower child along the axis first
        stack.push_back({first + 1u, t.lo + best_cut, t.hi, t.link});
        stack.push_back({first, t.lo, t.lo + best_cut, first + 1u});
    }
    (void)hipSetDevice(c->device);
    RZ_HIP(c, c->shadow_nodes64.resize(rec.size()));
    RZ_HIP(c, c->shadow_order.resize(n));
    RZ_HIP(c, hipStreamSynchronize(c->stream));
    RZ_HIP(c, hipMemcpy(c->shadow_nodes64.ptr, rec.data(), rec.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    RZ_HIP(c, hipMemcpy(c->shadow_order.ptr, order.data(), size_t(n) * sizeof(uint32_t), hipMemcpyHostToDevice));
    d.shadow_nodes64 = reinterpret_cast<const float4*>(c->shadow_nodes64.ptr), d.shadow_order = c->shadow_order.ptr, d.shadow_root = 0u;
    return HIPRZ_OK;
}

int restart_after_geometry_change(hiprz_ctx* c) {
    c->reset_pending = true;  // the world changed: accumulation restarts (cpu_engine_renderer.cpp:108-112), for every camera
    for (auto& f : c->parked) f.reset_pending = true;
    return HIPRZ_OK;
}
}  // namespace

int hiprz_update_triangles(hiprz_ctx* c, uint32_t first, uint32_t n, const hiprz_tri* tris, const hiprz_tri_attr* attrs) {
    if (!c) return HIPRZ_ERR_INVALID;
    RZ_FANOUT_OTHER_DEVICES(c, hiprz_update_triangles(p, first, n, tris, attrs));
    struct Share {
        hiprz_ctx* c;
        ~Share() { share_scene_with_streams(c); }
    } share{c};
    for (hiprz_ctx* p : c->peers)
        if (p->device == c->device) (void)hipStreamSynchronize(p->stream);
    if (!c->have_scene || c->scene_tree != HIPRZ_TREE_DEVICE) return fail(c, HIPRZ_ERR_STATE, "update_triangles: the scene was not uploaded under HIPRZ_TREE_DEVICE");
    if (n == 0u) return HIPRZ_OK;
    if (!tris || !attrs || uint64_t(first) + n > c->n_tris) return fail(c, HIPRZ_ERR_INVALID, "update_triangles: range outside the uploaded triangles");
    for (uint32_t k = 0; k < n; ++k)  // the walks divide by nothing here, but the shading indexes material slots
        if ((tris[k].material_flags & HIPRZ_TRI_MATERIAL_MASK) > 0xFFFFFFu) return fail(c, HIPRZ_ERR_INVALID, "update_triangles: bad material id");
    (void)hipSetDevice(c->device);
    RZ_HIP(c, hipStreamSynchronize(c->stream));
    const int rc = device_update_triangles(c, first, n, tris, attrs);
    if (rc != HIPRZ_OK) return scene_lost(c, rc);
    return restart_after_geometry_change(c);
}

int hiprz_rebuild_trees(hiprz_ctx* c, uint32_t tree) {
    if (!c) return HIPRZ_ERR_INVALID;
    RZ_FANOUT_OTHER_DEVICES(c, hiprz_rebuild_trees(p, tree));
    struct Share {
        hiprz_ctx* c;
        ~Share() { share_scene_with_streams(c); }
    } share{c};
    for (hiprz_ctx* p : c->peers)
        if (p->device == c->device) (void)hipStreamSynchronize(p->stream);
    if (!c->have_scene || c->scene_tree != HIPRZ_TREE_DEVICE) return fail(c, HIPRZ_ERR_STATE, "rebuild_trees: the scene's trees were not built on the device");
    if (tree != HIPRZ_TREE_DEVICE && tree != HIPRZ_TREE_DEVICE_SAH) return fail(c, HIPRZ_ERR_INVALID, "rebuild_trees: HIPRZ_TREE_DEVICE or HIPRZ_TREE_DEVICE_SAH");
    (void)hipSetDevice(c->device);
    RZ_HIP(c, hipStreamSynchronize(c->stream));
    invalidate_graphs(c);
    c->build_sah = tree == HIPRZ_TREE_DEVICE_SAH;
    // the meshes' bounds as they are now: the box of every root (exact after a refit)
    std::vector<DeviceMesh> meshes = c->device_meshes;
    for (DeviceMesh& m : meshes) {
        const uint32_t root = m.region != RZ_END ? m.region : m.leaf_slot;
        if (root == RZ_END || m.n_tris == 0u) continue;
        float rec[8];
        RZ_HIP(c, hipMemcpy(rec, c->dev_nodes.ptr + size_t(root) * sizeof(hiprz_node), sizeof rec, hipMemcpyDeviceToHost));
        m.bb_min[0] = rec[0], m.bb_max[0] = rec[1], m.bb_min[1] = rec[2], m.bb_max[1] = rec[3], m.bb_min[2] = rec[4], m.bb_max[2] = rec[5];
    }
    const std::vector<uint32_t> instance_mesh = c->instance_mesh;
    const int rc = device_build_mesh_trees(c, meshes, instance_mesh, !std::getenv("HIPRZ_TRUST_DEVICE_TREES"));
    if (rc != HIPRZ_OK) return scene_lost(c, rc);  // nodes, links and the triangle order were being rewritten in place
    for (size_t i = 0; i < c->device_instances.size(); ++i)
        if (c->instance_mesh[i] != RZ_END && c->device_meshes[c->instance_mesh[i]].region != RZ_END)
            c->device_instances[i].blas_root = c->device_meshes[c->instance_mesh[i]].region;
    uint32_t emitted = c->world_slots;
    for (const auto& m : c->device_meshes) emitted += m.n_slots;
    c->n_nodes = emitted;
    resolve_pipeline(c);
    return restart_after_geometry_change(c);
}

int hiprz_update_instances(hiprz_ctx* c, const hiprz_instance* instances, uint32_t n) {
    if (!c) return HIPRZ_ERR_INVALID;
    RZ_FANOUT_OTHER_DEVICES(c, hiprz_update_instances(p, instances, n));
    struct Share {
        hiprz_ctx* c;
        ~Share() { share_scene_with_streams(c); }
    } share{c};
    for (hiprz_ctx* p : c->peers)
        if (p->device == c->device) (void)hipStreamSynchronize(p->stream);
    if (!c->have_scene || c->scene_tree != HIPRZ_TREE_DEVICE) return fail(c, HIPRZ_ERR_STATE, "update_instances: the scene was not uploaded under HIPRZ_TREE_DEVICE");
    if (!instances || n != c->dscene.n_instances || n != c->device_instances.size()) return fail(c, HIPRZ_ERR_INVALID, "update_instances: the scene was uploaded with " + std::to_string(c->dscene.n_instances) + " instances");
    bool fast_div = c->dscene.fast_div != 0u;
    auto coord_ok = [](float x) {
        uint32_t b;
        std::memcpy(&b, &x, 4);
        const uint32_t e = (b >> 23) & 0xFFu;
        return (b & 0x7FFFFFFFu) == 0u || (e >= 127u - 60u && e < 127u + 40u);
    };
    for (uint32_t i = 0; i < n; ++i) {
        hiprz_instance& d = c->device_instances[i];  // keeps blas_root (the device-built root), the material table and the padding flags
        const hiprz_instance& in = instances[i];
        std::memcpy(d.position, in.position, 12), std::memcpy(d.scale, in.scale, 12);
        std::memcpy(d.x_axis, in.x_axis, 12), std::memcpy(d.y_axis, in.y_axis, 12), std::memcpy(d.z_axis, in.z_axis, 12);
        d.pad0 = (in.scale[0] == 1.0f && in.scale[1] == 1.0f && in.scale[2] == 1.0f) ? 1u : 0u;
        const float v[6] = {in.bb_min[0], in.bb_max[0], in.bb_min[1], in.bb_max[1], in.bb_min[2], in.bb_max[2]};  // interleaved like nodes
        d.bb_min[0] = v[0], d.bb_min[1] = v[1], d.bb_min[2] = v[2];
        std::memcpy(&d.pad2, &v[3], 4);
        d.bb_max[0] = v[4], d.bb_max[1] = v[5], d.bb_max[2] = 0.0f;
        for (float x : v) fast_div = fast_div && coord_ok(x);
    }
    (void)hipSetDevice(c->device);
    RZ_HIP(c, hipStreamSynchronize(c->stream));
    RZ_HIP(c, hipMemcpy(c->hot.ptr + c->dscene.off_instances, c->device_instances.data(), sizeof(hiprz_instance) * n, hipMemcpyHostToDevice));
    if (!fast_div && c->dscene.fast_div) c->dscene.fast_div = 0u, invalidate_graphs(c);
    if (c->n_tlas_order) {
        const int rc = device_build_world_tree(c, !std::getenv("HIPRZ_TRUST_DEVICE_TREES"));
        if (rc != HIPRZ_OK) return scene_lost(c, rc);  // the new instance records and a world tree nobody proved are on the device
    }
    {
        const int src = build_shadow_world_tree(c, c->device_instances, c->dscene);  // the instances moved: the shadow rays' tree over them again
        if (src != HIPRZ_OK) return scene_lost(c, src);
    }
    return restart_after_geometry_change(c);
}

int hiprz_download_trees(hiprz_ctx* c, hiprz_node* nodes_out, uint32_t max_nodes, uint32_t* n_nodes_out, uint32_t* tlas_root_out, uint32_t* tlas_order_out,
                         uint32_t* blas_roots_out, uint32_t* tri_refpos_out) {
    if (!c) return HIPRZ_ERR_INVALID;
    if (!c->have_scene) return fail(c, HIPRZ_ERR_STATE, "download_trees before upload_scene");
    (void)hipSetDevice(c->device);
    RZ_HIP(c, hipStreamSynchronize(c->stream));
    const bool device_trees = c->scene_tree == HIPRZ_TREE_DEVICE;
    const uint32_t n_nodes = device_trees ? c->node_capacity : uint32_t((c->dscene.off_tlas_order - c->dscene.off_nodes) / sizeof(hiprz_node));
    if (n_nodes_out) *n_nodes_out = n_nodes;
    if (tlas_root_out) *tlas_root_out = c->dscene.tlas_root;
    if (nodes_out) {
        if (max_nodes < n_nodes) return fail(c, HIPRZ_ERR_INVALID, "download_trees: " + std::to_string(n_nodes) + " nodes");
        RZ_HIP(c, hipMemcpy(nodes_out, c->dscene.nodes, sizeof(hiprz_node) * n_nodes, hipMemcpyDeviceToHost));
        for (uint32_t k = 0; k < n_nodes; ++k) {  // the device keeps boxes interleaved: (min.x, max.x, min.y, max.y, min.z, max.z)
            hiprz_node& nd = nodes_out[k];
            const float v[6] = {nd.bb_min[0], nd.bb_min[1], nd.bb_min[2], nd.bb_max[0], nd.bb_max[1], nd.bb_max[2]};
            nd.bb_min[0] = v[0], nd.bb_max[0] = v[1], nd.bb_min[1] = v[2], nd.bb_max[1] = v[3], nd.bb_min[2] = v[4], nd.bb_max[2] = v[5];
        }
    }
    if (tlas_order_out && c->n_tlas_order) RZ_HIP(c, hipMemcpy(tlas_order_out, c->hot.ptr + c->dscene.off_tlas_order, 4u * c->n_tlas_order, hipMemcpyDeviceToHost));
    if (blas_roots_out && c->dscene.n_instances) {
        std::vector<hiprz_instance> inst(c->dscene.n_instances);
        RZ_HIP(c, hipMemcpy(inst.data(), c->hot.ptr + c->dscene.off_instances, sizeof(hiprz_instance) * inst.size(), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < inst.size(); ++i) blas_roots_out[i] = inst[i].blas_root;
    }
    if (tri_refpos_out && c->n_tris) {
        std::vector<hiprz_tri> tris(c->n_tris);
        RZ_HIP(c, hipMemcpy(tris.data(), c->hot.ptr + c->dscene.off_tris, sizeof(hiprz_tri) * tris.size(), hipMemcpyDeviceToHost));
        for (size_t t = 0; t < tris.size(); ++t) tri_refpos_out[t] = tris[t].pad0;
    }
    return HIPRZ_OK;
}

int hiprz_upload_camera(hiprz_ctx* c, const hiprz_camera* cam) {
    if (!c) return HIPRZ_ERR_INVALID;
    RZ_FANOUT(c, hiprz_upload_camera(p, cam));
    if (!cam) return fail(c, HIPRZ_ERR_INVALID, "upload_camera: camera is null");
    if (cam->width == 0 || cam->height == 0 || cam->width > 32768u || cam->height > 32768u)
        return fail(c, HIPRZ_ERR_INVALID, "upload_camera: resolution must be 1..32768");
    StageTimer timer;
    (void)hipSetDevice(c->device);
    const bool resized = !c->have_camera || cam->width != c->camera.width || cam->height != c->camera.height;
    if (!c->have_camera || std::memcmp(&c->camera, cam, sizeof(hiprz_camera)) != 0) c->graph_valid = false;
    c->camera = *cam;
    DCamera& d = c->dcamera;
    std::memcpy(d.position, cam->position, 12);
    std::memcpy(d.x_axis, cam->x_axis, 12);
    std::memcpy(d.y_axis, cam->y_axis, 12);
    std::memcpy(d.z_axis, cam->z_axis, 12);
    d.width = cam->width, d.height = cam->height;
    d.tan_half_fov = cam->tan_half_fov, d.aspect_ratio = cam->aspect_ratio;
    d.near_ = cam->near_far[0], d.far_ = cam->near_far[1];
    d.focal_distance = cam->focal_distance, d.aperture = cam->aperture, d.exposure_time = cam->exposure_time;
    c->have_camera = true;
    if (resized) {
        const int rc = allocate_frame(c);
        if (rc != HIPRZ_OK) return rc;
    }
    c->reset_pending = true;  // camera changed => context.reset (cpu_engine_renderer.cpp:108-112)
    c->timings.set("upload camera", timer.ms());
    return HIPRZ_OK;
}

int hiprz_set_config(hiprz_ctx* c, const hiprz_config* cfg) {
    if (!c) return HIPRZ_ERR_INVALID;
    if (!cfg) return fail(c, HIPRZ_ERR_INVALID, "set_config: config is null");
    for (size_t r = 0; r < c->peers.size(); ++r) {  // HIPRZ_SHARD_SAMPLES: part k of the context draws from the seed stream seed + k
        hiprz_ctx* p = c->peers[r];
        hiprz_config part = *cfg;
        if (c->shard_mode == HIPRZ_SHARD_SAMPLES) part.seed += uint32_t(r) + 1u;
        const int rz_rc = hiprz_set_config(p, &part);
        if (rz_rc != HIPRZ_OK) return fail(c, rz_rc, "device " + std::to_string(p->device) + ": " + p->error);
    }
    if (cfg->max_depth == 0 || cfg->max_depth > 254u) return fail(c, HIPRZ_ERR_INVALID, "max_depth must be 1..254 (u8, 255 = path ended)");
    // The CPU kernel divides by sample_count/light_count and yields NaN for 0 samples
    // (cpu_engine_kernel.cpp:742-743, 789-790); the CUDA backend clamps to >= 1 (cuda_kernel_data.cu:23-31).
    if (cfg->spot_samples == 0 || cfg->direct_samples == 0 || cfg->spot_samples > 255u || cfg->direct_samples > 255u)
        return fail(c, HIPRZ_ERR_INVALID, "light sample counts must be 1..255");
    assign_setting(c, c->config, *cfg);
    return HIPRZ_OK;
}

namespace {
int reset_all_cameras(hiprz_ctx* c) {
    for (hiprz_ctx* p : c->peers) (void)reset_all_cameras(p);
    c->reset_pending = true;
    for (auto& f : c->parked) f.reset_pending = true;
    return HIPRZ_OK;
}
int set_shard_one(hiprz_ctx* c, uint32_t rank, uint32_t world) {
    const bool changed = rank != c->rank || world != c->world;
    c->rank = rank, c->world = world;
    if (!changed) return HIPRZ_OK;
    invalidate_graphs(c);
    (void)hipSetDevice(c->device);
    const uint32_t active = c->active_camera;
    for (uint32_t k = 0; k < uint32_t(c->parked.size()); ++k) {  // every camera's frame is re-tiled for the new shard
        select_camera_one(c, k);
        if (c->have_camera) {
            const int rc = allocate_frame(c);
            if (rc != HIPRZ_OK) return rc;
            c->reset_pending = true;
        }
    }
    select_camera_one(c, active);
    return HIPRZ_OK;
}
}  // namespace

int hiprz_set_shard(hiprz_ctx* c, uint32_t rank, uint32_t world) {
    if (!c) return HIPRZ_ERR_INVALID;
    if (world == 0 || rank >= world) return fail(c, HIPRZ_ERR_INVALID, "set_shard: need rank < world");
    // a multi-device context splits ITS shard once more over its devices: device r of n renders shard rank * n + r of world * n
    // (HIPRZ_SHARD_SAMPLES: every part renders the whole of the context's shard, on its own seed stream)
    const bool samples = c->shard_mode == HIPRZ_SHARD_SAMPLES;
    const uint32_t n = samples ? 1u : uint32_t(c->peers.size()) + 1u;
    c->user_rank = rank, c->user_world = world;
    for (uint32_t r = 1; r <= uint32_t(c->peers.size()); ++r) {
        const int rc = set_shard_one(c->peers[r - 1u], samples ? rank : rank * n + r, world * n);
        if (rc != HIPRZ_OK) return fail(c, rc, "device " + std::to_string(c->peers[r - 1u]->device) + ": " + c->peers[r - 1u]->error);
    }
    return set_shard_one(c, rank * n, world * n);
}

int hiprz_set_shard_mode(hiprz_ctx* c, uint32_t mode) {
    if (!c) return HIPRZ_ERR_INVALID;
    if (mode > HIPRZ_SHARD_SAMPLES) return fail(c, HIPRZ_ERR_INVALID, "set_shard_mode: HIPRZ_SHARD_TILES or HIPRZ_SHARD_SAMPLES");
    if (mode == c->shard_mode) return HIPRZ_OK;
    c->shard_mode = mode;
    if (c->peers.empty()) return HIPRZ_OK;  // one part: both modes are the same thing
    // the parts' shards and seed streams follow the mode; whatever was accumulated under the other one does not mix with it
    int rc = hiprz_set_shard(c, c->user_rank, c->user_world);
    if (rc != HIPRZ_OK) return rc;
    const hiprz_config cfg = c->config;
    rc = hiprz_set_config(c, &cfg);
    if (rc != HIPRZ_OK) return rc;
    invalidate_graphs(c);
    for (hiprz_ctx* p : c->peers) invalidate_graphs(p);
    return reset_all_cameras(c);
}

int hiprz_shard_mode(hiprz_ctx* c, uint32_t* mode_out) {
    if (!c || !mode_out) return HIPRZ_ERR_INVALID;
    *mode_out = c->shard_mode;
    return HIPRZ_OK;
}

int hiprz_set_traversal_mode(hiprz_ctx* c, int mode) {
    if (!c) return HIPRZ_ERR_INVALID;
    RZ_FANOUT(c, hiprz_set_traversal_mode(p, mode));
    if (mode < -1 || mode == 0 || mode > 3) return fail(c, HIPRZ_ERR_INVALID, "traversal mode: -1 = per scene, 1 = LDS stack, 2 = workgroup-binned, 3 = skip links (single-wave workgroups)");
    assign_setting(c, c->traversal_mode, mode);
    resolve_pipeline(c);
    return HIPRZ_OK;
}

int hiprz_set_mode(hiprz_ctx* c, uint32_t compat_flags) {
    if (!c) return HIPRZ_ERR_INVALID;
    RZ_FANOUT(c, hiprz_set_mode(p, compat_flags));
    if (compat_flags & ~HIPRZ_MODE_CUDA_COMPAT) return fail(c, HIPRZ_ERR_INVALID, "set_mode: unknown HIPRZ_COMPAT_* flag");
    if (compat_flags != c->mode_flags) {
        const bool integrator_changed = ((compat_flags ^ c->mode_flags) & kIntegratorFlags) != 0u;
        c->mode_flags = compat_flags;
        if (integrator_changed) {
            invalidate_graphs(c);
            c->reset_pending = true;  // another integrator: what has been accumulated does not mix with it
            for (auto& f : c->parked) f.reset_pending = true;
            resolve_pipeline(c);
        }
    }
    return HIPRZ_OK;
}

int hiprz_set_temporal_blend(hiprz_ctx* c, float blend) {
    if (!c) return HIPRZ_ERR_INVALID;
    RZ_FANOUT(c, hiprz_set_temporal_blend(p, blend));
    c->temporal_blend = std::min(std::max(blend, 0.0f), 1.0f);  // camera.cpp:154-156
    return HIPRZ_OK;
}

int hiprz_set_tree(hiprz_ctx* c, uint32_t tree) {
    if (!c) return HIPRZ_ERR_INVALID;
    RZ_FANOUT(c, hiprz_set_tree(p, tree));
    if (tree > HIPRZ_TREE_AUTO) return fail(c, HIPRZ_ERR_INVALID, "set_tree: HIPRZ_TREE_REFERENCE, _SAH, _DEVICE, _DEVICE_SAH or _AUTO");
    c->tree_mode = tree == HIPRZ_TREE_DEVICE_SAH ? HIPRZ_TREE_DEVICE : tree;  // one kind of scene (device-built, refittable), two builders
    c->device_sah = tree == HIPRZ_TREE_DEVICE_SAH;
    return HIPRZ_OK;
}

int hiprz_tree(hiprz_ctx* c, uint32_t* tree_out) {
    if (!c || !tree_out) return HIPRZ_ERR_INVALID;
    if (!c->have_scene) return fail(c, HIPRZ_ERR_STATE, "tree: no scene uploaded");
    *tree_out = c->scene_tree == HIPRZ_TREE_DEVICE && c->build_sah ? HIPRZ_TREE_DEVICE_SAH : c->scene_tree;
    return HIPRZ_OK;
}

int hiprz_set_walk_order(hiprz_ctx* c, int order) {
    if (!c) return HIPRZ_ERR_INVALID;
    RZ_FANOUT(c, hiprz_set_walk_order(p, order));
    if (order < 0 || order > 2) return fail(c, HIPRZ_ERR_INVALID, "walk order: 0 = the reference's child order, 1 = front-to-back, 2 = front-to-back also in counted renders");
    assign_setting(c, c->walk_order, order);
    return HIPRZ_OK;
}

int hiprz_set_lds_scene(hiprz_ctx* c, int mode) {
    if (!c) return HIPRZ_ERR_INVALID;
    RZ_FANOUT(c, hiprz_set_lds_scene(p, mode));
    if (mode < -1 || mode > 1) return fail(c, HIPRZ_ERR_INVALID, "lds scene: -1 auto, 0 off, 1 on");
    assign_setting(c, c->lds_scene_override, mode);
    resolve_pipeline(c);
    return HIPRZ_OK;
}

int hiprz_set_pipeline(hiprz_ctx* c, int pipeline) {
    if (!c) return HIPRZ_ERR_INVALID;
    RZ_FANOUT(c, hiprz_set_pipeline(p, pipeline));
    if (pipeline < -1 || pipeline > 2) return fail(c, HIPRZ_ERR_INVALID, "pipeline: -1 = per scene, 0 = fused pass kernel, 1 = trace kernel + shade kernel, 2 = resident (one launch per batch of passes)");
    assign_setting(c, c->pipeline_setting, pipeline);
    resolve_pipeline(c);
    return HIPRZ_OK;
}

int hiprz_pipeline(hiprz_ctx* c, int* out) {
    if (!c || !out) return HIPRZ_ERR_INVALID;
    *out = c->pipeline;
    return HIPRZ_OK;
}

int hiprz_traversal_mode(hiprz_ctx* c, int* out) {
    if (!c || !out) return HIPRZ_ERR_INVALID;
    *out = effective_mode(c);
    return HIPRZ_OK;
}

int hiprz_set_ray_sort(hiprz_ctx* c, int mode) {
    if (!c) return HIPRZ_ERR_INVALID;
    RZ_FANOUT(c, hiprz_set_ray_sort(p, mode));
    if (mode < -1 || mode > 1) return fail(c, HIPRZ_ERR_INVALID, "ray sort: -1 auto, 0 off, 1 on");
    assign_setting(c, c->sort_rays, mode);
    return HIPRZ_OK;
}

int hiprz_set_xcd_swizzle(hiprz_ctx* c, int enabled) {
    if (!c) return HIPRZ_ERR_INVALID;
    RZ_FANOUT(c, hiprz_set_xcd_swizzle(p, enabled));
    assign_setting(c, c->xcd_swizzle, enabled != 0);
    return HIPRZ_OK;
}

int hiprz_set_graph(hiprz_ctx* c, int enabled) {
    if (!c) return HIPRZ_ERR_INVALID;
    RZ_FANOUT(c, hiprz_set_graph(p, enabled));
    c->use_graph = enabled != 0;
    return HIPRZ_OK;
}

int hiprz_graph_captures(hiprz_ctx* c, uint32_t* out) {
    if (!c || !out) return HIPRZ_ERR_INVALID;
    *out = c->graph_captures;
    return HIPRZ_OK;
}

int hiprz_reset(hiprz_ctx* c) {
    if (!c) return HIPRZ_ERR_INVALID;
    RZ_FANOUT(c, hiprz_reset(p));
    c->reset_pending = true;
    return HIPRZ_OK;
}

int hiprz_render(hiprz_ctx* c, uint32_t n_passes) {
    if (!c) return HIPRZ_ERR_INVALID;
    if (!c->peers.empty() && n_passes) {
        const int rc = assemble_history(c);
        if (rc != HIPRZ_OK) return rc;
    }
    RZ_FANOUT(c, hiprz_render(p, n_passes));
    (void)hipSetDevice(c->device);
    return render_impl(c, n_passes, false);
}

int hiprz_render_counted(hiprz_ctx* c, uint32_t n_passes, hiprz_counters* out) {
    if (!c) return HIPRZ_ERR_INVALID;
    if (!out) return fail(c, HIPRZ_ERR_INVALID, "render_counted: out is null");
    hiprz_counters peers_total{};
    for (hiprz_ctx* p : c->peers) {
        hiprz_counters part{};
        const int rz_rc = hiprz_render_counted(p, n_passes, &part);
        if (rz_rc != HIPRZ_OK) return fail(c, rz_rc, "device " + std::to_string(p->device) + ": " + p->error);
        uint64_t* t = reinterpret_cast<uint64_t*>(&peers_total);
        const uint64_t* q = reinterpret_cast<const uint64_t*>(&part);
        for (size_t k = 0; k < sizeof(hiprz_counters) / sizeof(uint64_t); ++k) t[k] += q[k];
    }
    (void)hipSetDevice(c->device);
    RZ_HIP(c, hipMemsetAsync(c->counters_dev.ptr, 0, 16 * sizeof(unsigned long long), c->stream));
    const int rc = render_impl(c, n_passes, true);
    if (rc != HIPRZ_OK) return rc;
    unsigned long long v[10];
    RZ_HIP(c, hipMemcpyAsync(v, c->counters_dev.ptr, sizeof v, hipMemcpyDeviceToHost, c->stream));
    RZ_HIP(c, hipStreamSynchronize(c->stream));
    out->segments = v[0], out->box_tests = v[1], out->tri_tests = v[2], out->hits = v[3];
    out->shadow_rays = v[4], out->light_samples = v[5], out->texel_fetches = v[6], out->finished = v[7];
    out->shadow_box_tests = v[8], out->shadow_tri_tests = v[9];
    {
        uint64_t* t = reinterpret_cast<uint64_t*>(out);
        const uint64_t* q = reinterpret_cast<const uint64_t*>(&peers_total);
        for (size_t k = 0; k < sizeof(hiprz_counters) / sizeof(uint64_t); ++k) t[k] += q[k];
    }
    return HIPRZ_OK;
}

int hiprz_tonemap(hiprz_ctx* c) {
    if (!c) return HIPRZ_ERR_INVALID;
    if (samples_head(c)) {  // the tone map of the SUM of the parts' accumulators, into the head's pixels
        if (!c->have_camera) return fail(c, HIPRZ_ERR_STATE, "tonemap before camera upload");
        (void)hipSetDevice(c->device);
        const uint32_t n = c->n_local_tiles * 256u;
        if (!n) return HIPRZ_OK;
        RZ_HIP(c, c->sum_accum.resize(n));
        const int rc = sum_parts(c, c->sum_accum.ptr);
        if (rc != HIPRZ_OK) return rc;
        RZ_LAUNCH(rz_tonemap_tiles_kernel, dim3(c->n_local_tiles), dim3(256), 0, c->stream, c->sum_accum.ptr, c->rgba8.ptr, n, c->camera.aperture,
                           c->camera.exposure_time);
        RZ_HIP(c, hipGetLastError());
        return HIPRZ_OK;
    }
    RZ_FANOUT(c, hiprz_tonemap(p));
    if (!c->have_camera) return fail(c, HIPRZ_ERR_STATE, "tonemap before camera upload");
    (void)hipSetDevice(c->device);
    if (c->rgba8_valid && !c->reset_pending) return HIPRZ_OK;  // the resident kernel already wrote this frame's pixels
    const uint32_t n = c->n_local_tiles * 256u;
    if (n)
        RZ_LAUNCH(rz_tonemap_tiles_kernel, dim3(c->n_local_tiles), dim3(256), 0, c->stream, c->accum.ptr, c->rgba8.ptr, n,
                           c->camera.aperture, c->camera.exposure_time);
    RZ_HIP(c, hipGetLastError());
    return HIPRZ_OK;
}

int hiprz_sync(hiprz_ctx* c) {
    if (!c) return HIPRZ_ERR_INVALID;
    RZ_FANOUT(c, hiprz_sync(p));
    (void)hipSetDevice(c->device);
    RZ_HIP(c, hipStreamSynchronize(c->stream));
    return HIPRZ_OK;
}

int hiprz_read_rgba8(hiprz_ctx* c, uint8_t* dst, size_t bytes) {
    if (!c) return HIPRZ_ERR_INVALID;
    (void)hipSetDevice(c->device);
    return read_untiled<uint32_t>(c, c->rgba8.ptr, reinterpret_cast<uint32_t*>(dst), bytes, "read rgba8", [](hiprz_ctx* p) { return (const uint32_t*)p->rgba8.ptr; });
}
int hiprz_read_depth(hiprz_ctx* c, float* dst, size_t bytes) {
    if (!c) return HIPRZ_ERR_INVALID;
    (void)hipSetDevice(c->device);
    return read_untiled<float>(c, c->depth.ptr, dst, bytes, "read depth", [](hiprz_ctx* p) { return (const float*)p->depth.ptr; });
}
int hiprz_read_accum(hiprz_ctx* c, float* dst, size_t bytes) {
    if (!c) return HIPRZ_ERR_INVALID;
    (void)hipSetDevice(c->device);
    if (samples_head(c) && c->have_camera && c->n_local_tiles) {
        RZ_HIP(c, c->sum_accum.resize(size_t(c->n_local_tiles) * 256u));
        const int rc = sum_parts(c, c->sum_accum.ptr);
        if (rc != HIPRZ_OK) return rc;
        return read_untiled<float4>(c, c->sum_accum.ptr, reinterpret_cast<float4*>(dst), bytes, "read accum", [](hiprz_ctx* p) { return (const float4*)p->accum.ptr; });
    }
    return read_untiled<float4>(c, c->accum.ptr, reinterpret_cast<float4*>(dst), bytes, "read accum", [](hiprz_ctx* p) { return (const float4*)p->accum.ptr; });
}

int hiprz_read_state(hiprz_ctx* c, float* ray9, uint32_t* md2, size_t n_pixels) {
    if (!c) return HIPRZ_ERR_INVALID;
    if (!c->have_camera) return fail(c, HIPRZ_ERR_STATE, "readback before camera upload");
    const size_t n = size_t(c->camera.width) * c->camera.height;
    if (!ray9 || !md2 || n_pixels != n) return fail(c, HIPRZ_ERR_INVALID, "read_state: destination size mismatch");
    (void)hipSetDevice(c->device);
    RZ_HIP(c, c->state_ray.resize(9 * n));
    RZ_HIP(c, c->state_md.resize(2 * n));
    RZ_HIP(c, hipMemsetAsync(c->state_ray.ptr, 0, 9 * n * sizeof(float), c->stream));
    RZ_HIP(c, hipMemsetAsync(c->state_md.ptr, 0, 2 * n * sizeof(uint32_t), c->stream));
    if (c->n_local_tiles)
        RZ_LAUNCH(rz_untile_state_kernel, dim3(c->n_local_tiles), dim3(256), 0, c->stream, c->st0.ptr, c->st1.ptr,
                           c->st2.ptr, c->state_ray.ptr, c->state_md.ptr, c->camera.width, c->camera.height, c->tiles_x, c->rank,
                           c->world);
    for (hiprz_ctx* p : c->peers) {  // multi-device head: the peers' path state, one peer at a time through the gather buffer
        const size_t n_local = size_t(p->n_local_tiles) * 256u;
        if (!n_local || c->shard_mode == HIPRZ_SHARD_SAMPLES) continue;  // (sample mode: the parts walk different paths through the same pixels — part 0 answers)
        RZ_HIP(c, c->gather.resize(n_local * 40u));
        float4* g0 = reinterpret_cast<float4*>(c->gather.ptr);
        float4* g1 = g0 + n_local;
        float2* g2 = reinterpret_cast<float2*>(g1 + n_local);
        (void)hipSetDevice(p->device);
        RZ_HIP(c, hipEventRecord(p->peer_done, p->stream));
        (void)hipSetDevice(c->device);
        RZ_HIP(c, hipStreamWaitEvent(c->stream, p->peer_done, 0));
        RZ_HIP(c, hipMemcpyPeerAsync(g0, c->device, p->st0.ptr, p->device, n_local * 16u, c->stream));
        RZ_HIP(c, hipMemcpyPeerAsync(g1, c->device, p->st1.ptr, p->device, n_local * 16u, c->stream));
        RZ_HIP(c, hipMemcpyPeerAsync(g2, c->device, p->st2.ptr, p->device, n_local * 8u, c->stream));
        RZ_LAUNCH(rz_untile_state_kernel, dim3(p->n_local_tiles), dim3(256), 0, c->stream, g0, g1, g2, c->state_ray.ptr, c->state_md.ptr,
                           c->camera.width, c->camera.height, c->tiles_x, p->rank, p->world);
    }
    RZ_HIP(c, hipMemcpyAsync(ray9, c->state_ray.ptr, 9 * n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    RZ_HIP(c, hipMemcpyAsync(md2, c->state_md.ptr, 2 * n * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    RZ_HIP(c, hipStreamSynchronize(c->stream));
    return HIPRZ_OK;
}

int hiprz_ray_count(hiprz_ctx* c, uint64_t* out) {
    if (!c || !out) return HIPRZ_ERR_INVALID;
    *out = c->ray_count;
    for (hiprz_ctx* p : c->peers) *out += p->ray_count;
    return HIPRZ_OK;
}
int hiprz_pass_count(hiprz_ctx* c, uint32_t* out) {
    if (!c || !out) return HIPRZ_ERR_INVALID;
    *out = c->passes;
    return HIPRZ_OK;
}

int hiprz_local_pixel_capacity(hiprz_ctx* c, size_t* out) {
    if (!c || !out) return HIPRZ_ERR_INVALID;
    *out = c->peers.empty() || c->shard_mode == HIPRZ_SHARD_SAMPLES ? size_t(c->n_local_tiles) * 256u : part_capacity(c) * (c->peers.size() + 1u);
    return HIPRZ_OK;
}
int hiprz_export_accum_tiles(hiprz_ctx* c, void* dst_device, size_t bytes) {
    if (!c) return HIPRZ_ERR_INVALID;
    if (samples_head(c)) {  // the parts' sum, straight into the caller's buffer
        if (!dst_device || bytes < size_t(c->n_local_tiles) * 256u * sizeof(float4)) return fail(c, HIPRZ_ERR_INVALID, "export_accum_tiles: destination too small");
        return sum_parts(c, static_cast<float4*>(dst_device));
    }
    return export_tiles<float4>(c, dst_device, bytes, "export_accum_tiles", [](hiprz_ctx* x) { return (const float4*)x->accum.ptr; });
}
int hiprz_export_rgba8_tiles(hiprz_ctx* c, void* dst_device, size_t bytes) {
    if (!c) return HIPRZ_ERR_INVALID;
    return export_tiles<uint32_t>(c, dst_device, bytes, "export_rgba8_tiles", [](hiprz_ctx* x) { return (const uint32_t*)x->rgba8.ptr; });
}
int hiprz_untile_rgba8(hiprz_ctx* c, const void* src_tiles, uint32_t rank, uint32_t world, void* dst_image) {
    if (!c) return HIPRZ_ERR_INVALID;
    if (!c->have_camera) return fail(c, HIPRZ_ERR_STATE, "untile before camera upload");
    if (!src_tiles || !dst_image || world == 0 || rank >= world) return fail(c, HIPRZ_ERR_INVALID, "untile_rgba8: bad arguments");
    (void)hipSetDevice(c->device);
    const uint32_t n_local = shard_local_tiles(c->tiles_x, c->tiles_y, rank, world);
    if (n_local)
        RZ_LAUNCH((rz_untile_kernel<uint32_t>), dim3(n_local), dim3(256), 0, c->stream,
                           reinterpret_cast<const uint32_t*>(src_tiles), reinterpret_cast<uint32_t*>(dst_image), c->camera.width,
                           c->camera.height, c->tiles_x, rank, world);
    RZ_HIP(c, hipGetLastError());
    return HIPRZ_OK;
}
int hiprz_untile_gathered(hiprz_ctx* c, const void* src_parts, uint32_t world, size_t part_stride_bytes, uint32_t element_bytes,
                          void* dst_image, void* stream) {
    if (!c) return HIPRZ_ERR_INVALID;
    if (!c->have_camera) return fail(c, HIPRZ_ERR_STATE, "untile before camera upload");
    if (!src_parts || !dst_image || world == 0 || world > 65535u || (element_bytes != 4u && element_bytes != 16u) || part_stride_bytes % element_bytes)
        return fail(c, HIPRZ_ERR_INVALID, "untile_gathered: bad arguments");
    (void)hipSetDevice(c->device);
    const uint32_t n_tiles = c->tiles_x * c->tiles_y;
    const uint32_t per_rank = shard_local_tiles(c->tiles_x, c->tiles_y, 0u, world);
    if (part_stride_bytes < size_t(per_rank) * 256u * element_bytes) return fail(c, HIPRZ_ERR_INVALID, "untile_gathered: part stride smaller than a shard");
    hipStream_t st = stream ? static_cast<hipStream_t>(stream) : c->stream;
    if (n_tiles) {
        const dim3 grid(per_rank, world);
        if (element_bytes == 4u)
            RZ_LAUNCH((rz_untile_gathered_kernel<uint32_t>), grid, dim3(256), 0, st, reinterpret_cast<const uint32_t*>(src_parts),
                               part_stride_bytes / 4u, reinterpret_cast<uint32_t*>(dst_image), c->camera.width, c->camera.height, c->tiles_x, n_tiles, world, 0u);
        else
            RZ_LAUNCH((rz_untile_gathered_kernel<float4>), grid, dim3(256), 0, st, reinterpret_cast<const float4*>(src_parts),
                               part_stride_bytes / 16u, reinterpret_cast<float4*>(dst_image), c->camera.width, c->camera.height, c->tiles_x, n_tiles, world, 0u);
    }
    RZ_HIP(c, hipGetLastError());
    return HIPRZ_OK;
}
int hiprz_untile_accum(hiprz_ctx* c, const void* src_tiles, uint32_t rank, uint32_t world, void* dst_image) {
    if (!c) return HIPRZ_ERR_INVALID;
    if (!c->have_camera) return fail(c, HIPRZ_ERR_STATE, "untile before camera upload");
    if (!src_tiles || !dst_image || world == 0 || rank >= world) return fail(c, HIPRZ_ERR_INVALID, "untile_accum: bad arguments");
    (void)hipSetDevice(c->device);
    const uint32_t n_local = shard_local_tiles(c->tiles_x, c->tiles_y, rank, world);
    if (n_local)
        RZ_LAUNCH((rz_untile_kernel<float4>), dim3(n_local), dim3(256), 0, c->stream,
                           reinterpret_cast<const float4*>(src_tiles), reinterpret_cast<float4*>(dst_image), c->camera.width,
                           c->camera.height, c->tiles_x, rank, world);
    RZ_HIP(c, hipGetLastError());
    return HIPRZ_OK;
}
int hiprz_tonemap_image(hiprz_ctx* c, const void* src_image, void* dst_rgba8) { return hiprz_tonemap_image_on(c, src_image, dst_rgba8, nullptr); }
int hiprz_tonemap_image_on(hiprz_ctx* c, const void* src_image, void* dst_rgba8, void* stream) {
    if (!c) return HIPRZ_ERR_INVALID;
    if (!c->have_camera) return fail(c, HIPRZ_ERR_STATE, "tonemap before camera upload");
    if (!src_image || !dst_rgba8) return fail(c, HIPRZ_ERR_INVALID, "tonemap_image: null pointer");
    (void)hipSetDevice(c->device);
    const uint32_t n = c->camera.width * c->camera.height;
    RZ_LAUNCH(rz_tonemap_image_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream ? static_cast<hipStream_t>(stream) : c->stream,
                       reinterpret_cast<const float4*>(src_image), reinterpret_cast<uint32_t*>(dst_rgba8), n, c->camera.aperture,
                       c->camera.exposure_time);
    RZ_HIP(c, hipGetLastError());
    return HIPRZ_OK;
}
void* hiprz_stream(hiprz_ctx* c) { return c ? reinterpret_cast<void*>(c->stream) : nullptr; }

int hiprz_ray_cast(hiprz_ctx* c, uint32_t x, uint32_t y, hiprz_raycast* out) {
    if (!c) return HIPRZ_ERR_INVALID;
    if (!c->have_scene || !c->have_camera) return fail(c, HIPRZ_ERR_STATE, "ray cast before scene and camera upload");
    if (!out) return fail(c, HIPRZ_ERR_INVALID, "ray cast: null output");
    // Camera::rayCastPixel clamps (camera.cpp:159-165)
    if (x >= c->camera.width) x = c->camera.width - 1;
    if (y >= c->camera.height) y = c->camera.height - 1;
    (void)hipSetDevice(c->device);
    // depth of the pixel: only the shard that owns it can answer
    uint32_t owner, lt;
    shard_of_tile(x / 32u, y / 8u, c->tiles_x, c->world, owner, lt);
    *out = hiprz_raycast{-1, -1, -1, 0u};
    if (owner != c->rank) {
        for (hiprz_ctx* p : c->peers)
            if (p->world == c->world && owner == p->rank) {
                const int rc = hiprz_ray_cast(p, x, y, out);
                return rc == HIPRZ_OK ? rc : fail(c, rc, "device " + std::to_string(p->device) + ": " + p->error);
            }
        return HIPRZ_OK;
    }
    const uint32_t in_tile = ((x % 32u) / 8u) * 64u + (y % 8u) * 8u + (x % 8u);
    float depth = 0.0f;
    RZ_HIP(c, hipMemcpyAsync(&depth, c->depth.ptr + size_t(lt) * 256u + in_tile, sizeof(float), hipMemcpyDeviceToHost, c->stream));
    RZ_HIP(c, hipStreamSynchronize(c->stream));
    RZ_LAUNCH(rz_pick_kernel, dim3(1), dim3(1), 0, c->stream, c->dscene, c->dcamera, x, y, depth, c->pick_dev.ptr);
    int32_t out4[4] = {-1, -1, -1, 0};
    RZ_HIP(c, hipMemcpyAsync(out4, c->pick_dev.ptr, sizeof out4, hipMemcpyDeviceToHost, c->stream));
    RZ_HIP(c, hipStreamSynchronize(c->stream));
    out->instance = out4[0], out->material_slot = out4[1], out->material = out4[2], out->triangle = uint32_t(out4[3]);
    return HIPRZ_OK;
}

int hiprz_pick(hiprz_ctx* c, uint32_t x, uint32_t y, int32_t* instance_out, int32_t* material_out) {
    if (!c) return HIPRZ_ERR_INVALID;
    if (!instance_out || !material_out) return fail(c, HIPRZ_ERR_INVALID, "pick: null output");
    hiprz_raycast r;
    const int rc = hiprz_ray_cast(c, x, y, &r);
    *instance_out = rc == HIPRZ_OK ? r.instance : -1, *material_out = rc == HIPRZ_OK ? r.material : -1;
    return rc;
}

uint32_t hiprz_kernel_count(void) { return uint32_t(kernel_table().size()); }

int hiprz_selftest(hiprz_ctx* c, uint32_t cases_per_thread, uint32_t seed, uint64_t* mismatches, uint64_t* tested) {
    if (!c || !mismatches || !tested) return HIPRZ_ERR_INVALID;
    (void)hipSetDevice(c->device);
    RZ_HIP(c, hipMemsetAsync(c->counters_dev.ptr, 0, 8 * sizeof(unsigned long long), c->stream));
    RZ_LAUNCH(rz_selftest_div_kernel, dim3(1024), dim3(256), 0, c->stream, cases_per_thread, seed, c->counters_dev.ptr);
    unsigned long long v[3];
    RZ_HIP(c, hipMemcpyAsync(v, c->counters_dev.ptr, sizeof v, hipMemcpyDeviceToHost, c->stream));
    RZ_HIP(c, hipStreamSynchronize(c->stream));
    *mismatches = v[0], *tested = v[1];
    // how often the filtered box test has to fall back to the exact sequence on these (adversarial: half of them put a face on a range end) cases
    c->timings.set("selftest: box tests the filter left undecided, per million", double(v[2]) * 1.0e6 / double(262144ull * cases_per_thread));
    return HIPRZ_OK;
}

int hiprz_timings(hiprz_ctx* c, char* buf, size_t len) {
    if (!c || !buf || len == 0) return HIPRZ_ERR_INVALID;
    const std::string s = c->timings.str();
    std::snprintf(buf, len, "%s", s.c_str());
    return HIPRZ_OK;
}

int hiprz_time_kernels(hiprz_ctx* c, int enabled) {
    if (!c) return HIPRZ_ERR_INVALID;
    c->time_kernels = enabled != 0;  // timed batches are launched eagerly; a captured graph stays valid for the untimed ones
    return HIPRZ_OK;
}

int hiprz_kernel_breakdown_ms(hiprz_ctx* c, double* trace_ms, double* shade_ms, uint32_t* passes) {
    if (!c || !trace_ms || !shade_ms || !passes) return HIPRZ_ERR_INVALID;
    (void)hipSetDevice(c->device);
    RZ_HIP(c, hipStreamSynchronize(c->stream));
    *trace_ms = *shade_ms = 0.0;
    *passes = 0;
    if (resident_active(c)) {  // one launch for all the passes of the batch: reported as "trace"
        if (c->kernel_event_passes) {
            float a = 0;
            RZ_HIP(c, hipEventElapsedTime(&a, c->kernel_events[0], c->kernel_events[1]));
            *trace_ms = a;
            *passes = c->kernel_event_passes;
        }
        return HIPRZ_OK;
    }
    if (c->pipeline != 1) return HIPRZ_OK;
    for (uint32_t i = 0; i < c->kernel_event_passes; ++i) {
        float a = 0, b = 0;
        RZ_HIP(c, hipEventElapsedTime(&a, c->kernel_events[3 * i], c->kernel_events[3 * i + 1]));
        RZ_HIP(c, hipEventElapsedTime(&b, c->kernel_events[3 * i + 1], c->kernel_events[3 * i + 2]));
        *trace_ms += a, *shade_ms += b;
    }
    *passes = c->kernel_event_passes;
    return HIPRZ_OK;
}

int hiprz_kernel_time_ms(hiprz_ctx* c, double* total_ms, uint64_t* launches) {
    if (!c || !total_ms || !launches) return HIPRZ_ERR_INVALID;
    (void)hipSetDevice(c->device);
    RZ_HIP(c, hipStreamSynchronize(c->stream));
    double total = 0;
    uint64_t n = 0;
    for (size_t i = 0; i < c->pending_events.size(); ++i) {
        float ms = 0;
        RZ_HIP(c, hipEventElapsedTime(&ms, c->pending_events[i].first, c->pending_events[i].second));
        total += ms;
        n += c->pending_launches[i];
        c->event_pool.push_back(c->pending_events[i].first);
        c->event_pool.push_back(c->pending_events[i].second);
    }
    c->pending_events.clear();
    c->pending_launches.clear();
    *total_ms = total, *launches = n;
    if (n) c->timings.set("pass kernel (device)", total / double(n));
    return HIPRZ_OK;
}

}  // extern "C"
