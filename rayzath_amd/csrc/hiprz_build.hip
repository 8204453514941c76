// hiprz_build.hip — mesh trees built and refitted ON THE DEVICE.
//
// The reference rebuilds its trees on the host whenever a mesh changes (ComponentTreeNode::construct, component_container.hpp:259-363;
// Cuda::EngineCore re-mirrors them, cuda_engine_core.cu:58-60).  Here, with hiprz_set_tree(HIPRZ_TREE_DEVICE), every mesh tree of an
// uploaded scene is built by kernels — Morton codes of the triangle centroids, the radix sort of hiprz_sort.hip, the binary radix tree
// of Karras (2012) with one thread per inner node, boxes fitted bottom-up behind per-node arrival counters, subtrees of at most
// kLeafMax triangles collapsed into leaves — and emitted straight into the layout the walks read: 32-byte node records with the
// children of a node adjacent and the first child on the lower side of the split axis, the skip link under every ray octant, 64-byte
// walk records.  hiprz_update_triangles moves the vertices of a mesh and refits its boxes without touching the topology.
//
// A tree decides which boxes and triangles a ray meets, not what it hits: every triangle carries its position in the uploaded
// snapshot's order ("refpos") and equally distant hits are ranked by it, so frames are those of the reference trees bit for bit
// (tests/test_device_build_gpu.py).  The kernels follow the emitted links blindly, so the host proves on the downloaded tables that
// every walk terminates and that the leaves tile the mesh's triangles exactly before a device-built tree is used (validate_region).
#include <algorithm>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "hiprz_ctx.hpp"

namespace hiprz {
namespace {

constexpr uint32_t kLeafMax = 4u;  // one quad entry of the cooperative triangle phase
constexpr uint32_t kLeafBit = 0x80000000u;

struct BuildViews {  // device pointers of one mesh's build (all sized for the mesh's n triangles)
    const float4* tris;   // the scene's triangles (3 float4 each) in their order before the build, first triangle of the mesh at [0]
    const float4* attrs;  // ... and their shading records (6 float4 each; v2 and v3 ride in the padding words)
    uint32_t n;
    uint32_t* keys;
    uint32_t* sorted_keys;
    uint32_t* perm;         // sorted position -> triangle of the mesh
    uint32_t* left;         // [n - 1] child of an inner node: index | kLeafBit for a leaf (= sorted position)
    uint32_t* right;
    uint32_t* parent;       // [n - 1] parent of an inner node (root: RZ_END)
    uint32_t* leaf_parent;  // [n]
    uint32_t* first;        // [n - 1] first sorted position below the node
    uint32_t* count;        // [n - 1] triangles below the node
    uint32_t* visit;        // [n - 1] arrival counters of the bottom-up pass
    float* ibox;            // [n - 1][6] min.xyz max.xyz
    float* lbox;            // [n][6]
    uint32_t* survive;      // [n - 1] 1: stays an inner node
    uint32_t* rank;         // [n - 1] exclusive scan of survive; [n - 1] = total
    uint32_t* info;         // [n - 1] ptype | swapped << 2
};

RZ_DEV uint32_t expand10(uint32_t v) {  // 10 bits -> every third bit
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

__global__ void __launch_bounds__(256) rz_build_morton_kernel(BuildViews b, float3 lo, float3 scale) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= b.n) return;
    const float4 a = b.tris[3 * size_t(t)], e1 = b.tris[3 * size_t(t) + 1], e2 = b.tris[3 * size_t(t) + 2];
    const float third = 1.0f / 3.0f;
    const float cx = a.x + (e1.x + e2.x) * third, cy = a.y + (e1.y + e2.y) * third, cz = a.z + (e1.z + e2.z) * third;
    const uint32_t qx = uint32_t(fminf(fmaxf((cx - lo.x) * scale.x, 0.0f), 1023.0f));
    const uint32_t qy = uint32_t(fminf(fmaxf((cy - lo.y) * scale.y, 0.0f), 1023.0f));
    const uint32_t qz = uint32_t(fminf(fmaxf((cz - lo.z) * scale.z, 0.0f), 1023.0f));
    b.keys[t] = (expand10(qx) << 2) | (expand10(qy) << 1) | expand10(qz);
}

// length of the common prefix of the keys at sorted positions i and j (ties between equal keys are broken by the position itself)
RZ_DEV int common_prefix(const uint32_t* keys, uint32_t n, int i, int j) {
    if (j < 0 || j >= int(n)) return -1;
    const uint32_t a = keys[i], c = keys[j];
    return a == c ? 32 + __clz(uint32_t(i) ^ uint32_t(j)) : __clz(a ^ c);
}

// Karras 2012, "Maximizing parallelism in the construction of BVHs, octrees, and k-d trees": inner node i of the binary radix tree
// over the sorted keys, found independently of all others.
__global__ void __launch_bounds__(256) rz_build_radix_tree_kernel(BuildViews b) {
    const int i = int(blockIdx.x * 256u + threadIdx.x), n = int(b.n);
    if (i >= n - 1) return;
    const uint32_t* k = b.sorted_keys;
    const int d = common_prefix(k, b.n, i, i + 1) - common_prefix(k, b.n, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = common_prefix(k, b.n, i, i - d);
    int lmax = 2;
    while (common_prefix(k, b.n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (common_prefix(k, b.n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = common_prefix(k, b.n, i, j);
    int s = 0, t = l;
    do {
        t = (t + 1) / 2;
        if (common_prefix(k, b.n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    const int gamma = i + s * d + (d < 0 ? d : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    const bool left_leaf = lo == gamma, right_leaf = hi == gamma + 1;
    b.left[i] = left_leaf ? uint32_t(gamma) | kLeafBit : uint32_t(gamma);
    b.right[i] = right_leaf ? uint32_t(gamma + 1) | kLeafBit : uint32_t(gamma + 1);
    if (left_leaf) b.leaf_parent[gamma] = uint32_t(i);
    else b.parent[gamma] = uint32_t(i);
    if (right_leaf) b.leaf_parent[gamma + 1] = uint32_t(i);
    else b.parent[gamma + 1] = uint32_t(i);
    b.first[i] = uint32_t(lo);
    b.count[i] = uint32_t(hi - lo + 1);
    b.visit[i] = 0u;
    if (i == 0) b.parent[0] = RZ_END;
}

RZ_DEV void store_box(float* dst, const float* mn, const float* mx) {
    dst[0] = mn[0], dst[1] = mn[1], dst[2] = mn[2], dst[3] = mx[0], dst[4] = mx[1], dst[5] = mx[2];
}
// the triangle's vertices as uploaded: v1 in the intersection record, v2 and v3 in the padding words of the shading record
RZ_DEV void triangle_box(const float4* tris, const float4* attrs, size_t tri, float* mn, float* mx) {
    const float4 a = tris[3 * tri];
    const float4* at = attrs + 6 * tri;
    const float v2[3] = {at[0].w, at[1].w, at[2].w}, v3[3] = {at[3].w, at[5].z, at[5].w}, v1[3] = {a.x, a.y, a.z};
    for (int k = 0; k < 3; ++k) {
        mn[k] = fminf(fminf(v1[k], v2[k]), v3[k]);
        mx[k] = fmaxf(fmaxf(v1[k], v2[k]), v3[k]);
    }
}
// Boxes bottom-up: a thread starts at its leaf and climbs; at every inner node the first arrival stops, the second one — which knows
// that both children are complete — merges their boxes and goes on.  Boxes written by another compute unit are read behind an
// agent-scope fence on both sides (release before the counter, acquire after it).
__global__ void __launch_bounds__(256) rz_build_fit_kernel(BuildViews b) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= b.n) return;
    float mn[3], mx[3];
    triangle_box(b.tris, b.attrs, b.perm[t], mn, mx);
    store_box(b.lbox + 6 * size_t(t), mn, mx);
    if (b.n == 1u) return;
    uint32_t node = b.leaf_parent[t];
    while (node != RZ_END) {
        __threadfence();
        if (atomicAdd(&b.visit[node], 1u) == 0u) return;
        __threadfence();
        const uint32_t l = b.left[node], r = b.right[node];
        const float* lb = (l & kLeafBit) ? b.lbox + 6 * size_t(l & ~kLeafBit) : b.ibox + 6 * size_t(l);
        const float* rb = (r & kLeafBit) ? b.lbox + 6 * size_t(r & ~kLeafBit) : b.ibox + 6 * size_t(r);
        for (int k = 0; k < 3; ++k) mn[k] = fminf(lb[k], rb[k]), mx[k] = fmaxf(lb[3 + k], rb[3 + k]);
        store_box(b.ibox + 6 * size_t(node), mn, mx);
        node = b.parent[node];
    }
}

// which inner nodes stay inner nodes, on which axis their children are ordered, and whether the children swap places so that the
// FIRST child is the one on the lower side (what the walks' front-to-back rule assumes: bvh_tree_node.hpp:150-215 puts the centroids
// below the split plane into the first child)
__global__ void __launch_bounds__(256) rz_build_decide_kernel(BuildViews b) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i + 1u >= b.n) return;
    b.survive[i] = b.count[i] > kLeafMax ? 1u : 0u;
    const uint32_t l = b.left[i], r = b.right[i];
    const float* lb = (l & kLeafBit) ? b.lbox + 6 * size_t(l & ~kLeafBit) : b.ibox + 6 * size_t(l);
    const float* rb = (r & kLeafBit) ? b.lbox + 6 * size_t(r & ~kLeafBit) : b.ibox + 6 * size_t(r);
    int axis = 0;
    float best = -1.0f;
    bool swapped = false;
    for (int k = 0; k < 3; ++k) {
        const float cl = lb[k] + lb[3 + k], cr = rb[k] + rb[3 + k];
        if (fabsf(cl - cr) > best) best = fabsf(cl - cr), axis = k, swapped = cl > cr;
    }
    b.info[i] = uint32_t(2 - axis) | (swapped ? 4u : 0u);  // partition type X = 2, Y = 1, Z = 0 (bvh_tree_node.hpp:22-28)
}

// exclusive scan of `in[0..n)` into out[0..n), total into out[n]: one workgroup (the inputs are a few hundred thousand flags)
__global__ void __launch_bounds__(1024) rz_build_scan_kernel(const uint32_t* in, uint32_t* out, uint32_t n) {
    __shared__ uint32_t wave_sum[16];
    __shared__ uint32_t carry;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (tid == 0u) carry = 0u;
    __syncthreads();
    for (uint32_t base = 0u; base < n; base += 1024u) {
        const uint32_t i = base + tid, v = i < n ? in[i] : 0u;
        uint32_t incl = v;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = __shfl_up(incl, off);
            if (int(lane) >= off) incl += up;
        }
        if (lane == 63u) wave_sum[wave] = incl;
        __syncthreads();
        uint32_t before = carry;
        for (uint32_t w = 0; w < wave; ++w) before += wave_sum[w];
        if (i < n) out[i] = before + incl - v;
        __syncthreads();
        if (tid == 1023u) carry = before + incl;
        __syncthreads();
    }
    if (tid == 0u) out[n] = carry;
}

struct EmitViews {
    float4* nodes32;        // the scene's node records (2 float4 each, box interleaved min.x max.x min.y max.y | min.z max.z begin meta)
    uint32_t* slot_parent;  // [slots of the region] parent slot of an emitted node (root: RZ_END)
    uint32_t region;        // slot of the mesh's root; the children pair of surviving node i lives at region + 1 + 2 * rank[i]
    uint32_t tri_first;     // index of the mesh's first triangle in the scene's triangle array
};
RZ_DEV void write_node(const EmitViews& e, uint32_t slot, const float* box, uint32_t begin, uint32_t meta) {
    e.nodes32[2 * size_t(slot)] = make_float4(box[0], box[3], box[1], box[4]);
    e.nodes32[2 * size_t(slot) + 1] = make_float4(box[2], box[5], __uint_as_float(begin), __uint_as_float(meta));
}
__global__ void __launch_bounds__(256) rz_build_emit_kernel(BuildViews b, EmitViews e) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i + 1u >= b.n || !b.survive[i]) return;
    const uint32_t pair = e.region + 1u + 2u * b.rank[i];
    uint32_t my_slot = e.region;
    if (i != 0u) {
        const uint32_t p = b.parent[i], k = b.left[p] == i ? 0u : 1u;
        my_slot = e.region + 1u + 2u * b.rank[p] + (k ^ ((b.info[p] >> 2) & 1u));
    } else {
        write_node(e, e.region, b.ibox, pair, (b.info[0] & 3u) << HIPRZ_NODE_PTYPE_SHIFT);
        e.slot_parent[0] = RZ_END;
    }
    const uint32_t swapped = (b.info[i] >> 2) & 1u;
    for (uint32_t k = 0; k < 2u; ++k) {
        const uint32_t child = k == 0u ? b.left[i] : b.right[i], slot = pair + (k ^ swapped);
        if (child & kLeafBit) {
            const uint32_t t = child & ~kLeafBit;
            write_node(e, slot, b.lbox + 6 * size_t(t), e.tri_first + t, 1u | HIPRZ_NODE_LEAF);
        } else if (b.survive[child]) {
            write_node(e, slot, b.ibox + 6 * size_t(child), e.region + 1u + 2u * b.rank[child], (b.info[child] & 3u) << HIPRZ_NODE_PTYPE_SHIFT);
        } else {
            write_node(e, slot, b.ibox + 6 * size_t(child), e.tri_first + b.first[child], b.count[child] | HIPRZ_NODE_LEAF);
        }
        e.slot_parent[slot - e.region] = my_slot;
    }
}

// The skip link of every emitted node under every ray octant, found by climbing: the child a ray of octant o visits first links to
// its sibling, the other one inherits its parent's link (hiprz_api.hip: derive_tables does the same sweep top-down on the host).
// Writes the 64-byte walk records (node + 8 links) and the octant-0 links of the reference-order walks.
__global__ void __launch_bounds__(256) rz_build_links_kernel(EmitViews e, const uint32_t* n_pairs, uint32_t* nodes64, uint32_t* node_skip) {
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x, local = idx >> 3, o = idx & 7u;
    if (local >= 1u + 2u * *n_pairs) return;  // a root and its pairs of children
    const uint32_t slot = e.region + local;
    uint32_t x = slot, link = RZ_END;
    while (true) {
        const uint32_t p = e.slot_parent[x - e.region];
        if (p == RZ_END) break;
        const float4 pn = e.nodes32[2 * size_t(p) + 1];
        const uint32_t pbegin = __float_as_uint(pn.z), ptype = (__float_as_uint(pn.w) >> HIPRZ_NODE_PTYPE_SHIFT) & 3u;
        const uint32_t flip = (o >> ptype) & 1u;  // ptype 3 reads bit 3 = 0
        if (x == pbegin + flip) {
            link = pbegin + 1u - flip;
            break;
        }
        x = p;
    }
    uint32_t* rec = nodes64 + 16 * size_t(slot);
    rec[8u + o] = link;
    if (o == 0u) {
        const float4 n0 = e.nodes32[2 * size_t(slot)], n1 = e.nodes32[2 * size_t(slot) + 1];
        reinterpret_cast<float4*>(rec)[0] = n0, reinterpret_cast<float4*>(rec)[1] = n1;
        node_skip[slot] = link;
    }
}

// triangles and their shading records into the order of the new leaves
__global__ void __launch_bounds__(256) rz_build_permute_kernel(BuildViews b, float4* tris_out, float4* attrs_out) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= b.n) return;
    const size_t src = b.perm[t];
    for (int k = 0; k < 3; ++k) tris_out[3 * size_t(t) + k] = b.tris[3 * src + k];
    for (int k = 0; k < 6; ++k) attrs_out[6 * size_t(t) + k] = b.attrs[6 * src + k];
}

// position in the uploaded snapshot's order -> position on the device (hiprz_update_triangles addresses triangles as uploaded)
__global__ void __launch_bounds__(256) rz_build_refpos_kernel(const float4* tris, uint32_t n, uint32_t* ref_to_dev) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t < n) ref_to_dev[__float_as_uint(tris[3 * size_t(t) + 2].w)] = t;
}

// ---- refit ----
// new vertices of a run of triangles (addressed as uploaded) -> the device records: v1, the two edges (the same fp32 subtraction the
// upload performs), v2 / v3 in the shading record's padding, normals, texture coordinates and face normal as given
__global__ void __launch_bounds__(256) rz_refit_scatter_kernel(const hiprz_tri* tris, const hiprz_tri_attr* attrs, uint32_t first_ref, uint32_t n,
                                                               const uint32_t* ref_to_dev, float4* dtris, float4* dattrs) {
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k >= n) return;
    const hiprz_tri t = tris[k];
    const hiprz_tri_attr a = attrs[k];
    const size_t d = ref_to_dev ? ref_to_dev[first_ref + k] : first_ref + k;
    const float4 old2 = dtris[3 * d + 2];
    dtris[3 * d] = make_float4(t.v1[0], t.v1[1], t.v1[2], __uint_as_float(t.material_flags));
    dtris[3 * d + 1] = make_float4(t.v2[0] - t.v1[0], t.v2[1] - t.v1[1], t.v2[2] - t.v1[2], __uint_as_float(t.source_index));
    dtris[3 * d + 2] = make_float4(t.v3[0] - t.v1[0], t.v3[1] - t.v1[1], t.v3[2] - t.v1[2], old2.w);  // refpos stays
    dattrs[6 * d] = make_float4(a.n1[0], a.n1[1], a.n1[2], t.v2[0]);
    dattrs[6 * d + 1] = make_float4(a.n2[0], a.n2[1], a.n2[2], t.v2[1]);
    dattrs[6 * d + 2] = make_float4(a.n3[0], a.n3[1], a.n3[2], t.v2[2]);
    dattrs[6 * d + 3] = make_float4(a.face_normal[0], a.face_normal[1], a.face_normal[2], t.v3[0]);
    dattrs[6 * d + 4] = make_float4(a.t1[0], a.t1[1], a.t2[0], a.t2[1]);
    dattrs[6 * d + 5] = make_float4(a.t3[0], a.t3[1], t.v3[1], t.v3[2]);
}
// boxes of the emitted nodes bottom-up over the emitted topology: a thread per node; leaves compute their box from their triangles and
// climb behind arrival counters like the build's fit pass
__global__ void __launch_bounds__(256) rz_refit_kernel(EmitViews e, uint32_t n_slots, const float4* tris, const float4* attrs, uint32_t* visit, uint32_t* nodes64) {
    const uint32_t local = blockIdx.x * 256u + threadIdx.x;
    if (local >= n_slots) return;
    uint32_t slot = e.region + local;
    const float4 n1 = e.nodes32[2 * size_t(slot) + 1];
    const uint32_t meta = __float_as_uint(n1.w);
    if (!(meta & HIPRZ_NODE_LEAF)) return;
    const uint32_t begin = __float_as_uint(n1.z), count = meta & HIPRZ_NODE_COUNT_MASK;
    float mn[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
    for (uint32_t k = 0; k < count; ++k) {
        float a[3], c[3];
        triangle_box(tris, attrs, begin + k, a, c);
        for (int q = 0; q < 3; ++q) mn[q] = k ? fminf(mn[q], a[q]) : a[q], mx[q] = k ? fmaxf(mx[q], c[q]) : c[q];
    }
    while (true) {
        const float4 m1 = e.nodes32[2 * size_t(slot) + 1];
        const float4 b0 = make_float4(mn[0], mx[0], mn[1], mx[1]), b1 = make_float4(mn[2], mx[2], m1.z, m1.w);
        e.nodes32[2 * size_t(slot)] = b0, e.nodes32[2 * size_t(slot) + 1] = b1;
        reinterpret_cast<float4*>(nodes64 + 16 * size_t(slot))[0] = b0, reinterpret_cast<float4*>(nodes64 + 16 * size_t(slot))[1] = b1;
        const uint32_t p = e.slot_parent[slot - e.region];
        if (p == RZ_END) return;
        __threadfence();
        if (atomicAdd(&visit[p - e.region], 1u) == 0u) return;
        __threadfence();
        const uint32_t pair = __float_as_uint(e.nodes32[2 * size_t(p) + 1].z);
        const float4 l0 = e.nodes32[2 * size_t(pair)], l1 = e.nodes32[2 * size_t(pair) + 1], r0 = e.nodes32[2 * size_t(pair) + 2], r1 = e.nodes32[2 * size_t(pair) + 3];
        mn[0] = fminf(l0.x, r0.x), mx[0] = fmaxf(l0.y, r0.y), mn[1] = fminf(l0.z, r0.z), mx[1] = fmaxf(l0.w, r0.w);
        mn[2] = fminf(l1.x, r1.x), mx[2] = fmaxf(l1.y, r1.y);
        slot = p;
    }
}


// ---- the world tree on the device ----
// The tree over the instance boxes decides the ORDER in which a ray meets the instances, and a hit found in one instance rescales the
// ray's range through that instance's length factor (cpu_engine_kernel.cpp:308-330): the order is part of the arithmetic.  So the
// device rebuild is the reference's own top-down builder (TreeNode::construct, bvh_tree_node.hpp:117-215, as restated on the host in
// hiprz_host.cpp: FlatTreeBuilder) run by ONE thread, statement for statement: size split, running mean of the centroids, variance
// per axis, the partition of libstdc++'s std::partition — the same nodes, the same leaf order as the host call, bit for bit
// (tests/test_device_build_gpu.py).  A world has tens to thousands of instances; the kernel takes microseconds to a few milliseconds.
struct WorldViews {
    const float4* instances;  // device records (7 float4): [5] = (min.x, max.x, min.y, max.y), [6].xy = (min.z, max.z)
    const uint8_t* has_mesh;
    uint32_t n_instances;
    uint32_t* items;          // [n_instances] work array
    uint32_t* order;          // -> tlas_order
    float4* nodes32;
    uint32_t* slot_parent;    // relative to the region
    uint32_t region;
    uint32_t* n_pairs_out;    // (nodes - 1) / 2
    uint32_t* n_order_out;
};
struct WBox {
    float mn[3], mx[3];
};
RZ_DEV WBox instance_box(const WorldViews& w, uint32_t i) {
    const float4 a = w.instances[7 * size_t(i) + 5], b = w.instances[7 * size_t(i) + 6];
    return WBox{{a.x, a.z, b.x}, {a.y, a.w, b.y}};
}
RZ_DEV float wcentroid(const WBox& b, int a) { return (b.mn[a] + b.mx[a]) * 0.5f; }
// std::partition of libstdc++ for bidirectional iterators (what hiprz_host.cpp's std::partition calls run): returns the split point
template <typename Pred>
RZ_DEV uint32_t* partition_like_libstdcxx(uint32_t* first, uint32_t* last, Pred pred) {
    while (true) {
        while (true) {
            if (first == last) return first;
            if (pred(*first)) ++first;
            else break;
        }
        --last;
        while (true) {
            if (first == last) return first;
            if (!pred(*last)) --last;
            else break;
        }
        const uint32_t t = *first;
        *first = *last, *last = t;
        ++first;
    }
}
__global__ void rz_build_world_tree_kernel(WorldViews w) {
    if (threadIdx.x != 0u || blockIdx.x != 0u) return;
    constexpr uint32_t kLeaf = 4u, kRootLeaf = 8u, kMaxDepth = 31u;  // hiprz_build_world_tree: FlatTreeBuilder(boxes, 4, 8, ...)
    // ObjectContainerWithBVH::update (bvh.hpp:29-53): the box starts from instance 0 and grows by every instance that has a mesh
    uint32_t n_items = 0u;
    WBox root = w.n_instances ? instance_box(w, 0u) : WBox{{0, 0, 0}, {0, 0, 0}};
    for (uint32_t i = 0; i < w.n_instances; ++i)
        if (w.has_mesh[i]) {
            const WBox b = instance_box(w, i);
            for (int a = 0; a < 3; ++a) {
                if (root.mn[a] > b.mn[a]) root.mn[a] = b.mn[a];
                if (root.mx[a] < b.mx[a]) root.mx[a] = b.mx[a];
            }
            w.items[n_items++] = i;
        }
    struct Frame {
        uint32_t slot, begin, end, depth;
        WBox bb;
    };
    Frame stack[2 * kMaxDepth + 4];
    uint32_t sp = 0u, n_nodes = 1u, n_order = 0u;
    stack[sp++] = Frame{0u, 0u, n_items, 0u, root};
    w.slot_parent[0] = RZ_END;
    auto write = [&](uint32_t slot, const WBox& b, uint32_t begin, uint32_t meta) {
        w.nodes32[2 * size_t(w.region + slot)] = make_float4(b.mn[0], b.mx[0], b.mn[1], b.mx[1]);
        w.nodes32[2 * size_t(w.region + slot) + 1] = make_float4(b.mn[2], b.mx[2], __uint_as_float(begin), __uint_as_float(meta));
    };
    auto leaf = [&](uint32_t slot, uint32_t begin, uint32_t end) {
        WBox bb{{0, 0, 0}, {0, 0, 0}};
        const uint32_t first = n_order;
        for (uint32_t k = begin; k < end; ++k) {
            const WBox b = instance_box(w, w.items[k]);
            if (k == begin) bb = b;
            else
                for (int a = 0; a < 3; ++a) {
                    if (bb.mn[a] > b.mn[a]) bb.mn[a] = b.mn[a];
                    if (bb.mx[a] < b.mx[a]) bb.mx[a] = b.mx[a];
                }
            w.order[n_order++] = w.items[k];
        }
        write(slot, bb, first, (end - begin) | HIPRZ_NODE_LEAF);
    };
    while (sp) {
        const Frame f = stack[--sp];
        const uint32_t count = f.end - f.begin;
        if (f.depth > kMaxDepth || count <= kLeaf || (f.depth == 0u && count <= kRootLeaf)) {
            leaf(f.slot, f.begin, f.end);
            continue;
        }
        const float sx = f.bb.mx[0] - f.bb.mn[0], sy = f.bb.mx[1] - f.bb.mn[1], sz = f.bb.mx[2] - f.bb.mn[2];
        uint32_t* begin = w.items + f.begin;
        uint32_t* end = w.items + f.end;
        uint32_t* size_split = partition_like_libstdcxx(begin, end, [&](uint32_t i) {
            const WBox o = instance_box(w, i);
            return (o.mx[0] - o.mn[0]) < sx && (o.mx[1] - o.mn[1]) < sy && (o.mx[2] - o.mn[2]) < sz;
        });
        const uint32_t n_split = uint32_t(size_split - begin), n_large = uint32_t(end - size_split);
        auto inner = [&](uint32_t ptype, const WBox& b0, uint32_t s0, uint32_t e0, const WBox& b1, uint32_t s1, uint32_t e1) {
            const uint32_t c = n_nodes;
            n_nodes += 2u;
            write(f.slot, f.bb, w.region + c, ptype << HIPRZ_NODE_PTYPE_SHIFT);  // the box is fitted in the sweep below
            w.slot_parent[c] = w.region + f.slot, w.slot_parent[c + 1u] = w.region + f.slot;
            stack[sp++] = Frame{c + 1u, s1, e1, f.depth + 1u, b1};  // the first child's subtree is completed before the second's
            stack[sp++] = Frame{c, s0, e0, f.depth + 1u, b0};
        };
        if (n_split != 0u && n_large != 0u) {
            inner(3u, f.bb, f.begin, f.begin + n_split, f.bb, f.begin + n_split, f.end);
            continue;
        }
        if (n_split == 0u) {
            leaf(f.slot, f.begin + n_split, f.end);
            continue;
        }
        float spm[3] = {0.0f, 0.0f, 0.0f};  // running mean of the centroids
        for (uint32_t i = 0; i < n_split; ++i) {
            const WBox b = instance_box(w, begin[i]);
            for (int a = 0; a < 3; ++a) spm[a] += (wcentroid(b, a) - spm[a]) / float(i + 1u);
        }
        float var[3] = {0.0f, 0.0f, 0.0f};
        uint32_t below[3] = {0u, 0u, 0u};
        for (uint32_t i = 0; i < n_split; ++i) {
            const WBox b = instance_box(w, begin[i]);
            for (int a = 0; a < 3; ++a) {
                const float cc = wcentroid(b, a), d = cc - spm[a];
                var[a] += d * d;
                below[a] += uint32_t(cc < spm[a]);
            }
        }
        if (!below[0] && !below[1] && !below[2]) {
            leaf(f.slot, f.begin, f.begin + n_split);
            continue;
        }
        const float score[3] = {var[0] / float(n_split), var[1] / float(n_split), var[2] / float(n_split)};
        int axis = 2;
        if (score[0] >= score[1] && score[0] >= score[2] && below[0]) axis = 0;
        else if (score[1] >= score[0] && score[1] >= score[2] && below[1]) axis = 1;
        const float plane = spm[axis];
        uint32_t* mid = partition_like_libstdcxx(begin, size_split, [&](uint32_t i) { return wcentroid(instance_box(w, i), axis) < plane; });
        WBox b0 = f.bb, b1 = f.bb;
        // box_of(bb.mn, hi) / box_of(lo, bb.mx) with hi[axis] = lo[axis] = plane: std::min / std::max per axis
        b0.mx[axis] = plane, b1.mn[axis] = plane;
        for (int a = 0; a < 3; ++a) {
            const float p0 = b0.mn[a], q0 = b0.mx[a], p1 = b1.mn[a], q1 = b1.mx[a];
            b0.mn[a] = q0 < p0 ? q0 : p0, b0.mx[a] = p0 < q0 ? q0 : p0;
            b1.mn[a] = q1 < p1 ? q1 : p1, b1.mx[a] = p1 < q1 ? q1 : p1;
        }
        const uint32_t ptype = axis == 0 ? 2u : (axis == 1 ? 1u : 0u);
        inner(ptype, b0, f.begin, uint32_t(mid - w.items), b1, uint32_t(mid - w.items), f.begin + n_split);
    }
    // fitBoundingBox: children are opened after their parent, so their slots are higher — one descending sweep fits every inner box
    for (uint32_t slot = n_nodes; slot-- > 0u;) {
        const float4 m1 = w.nodes32[2 * size_t(w.region + slot) + 1];
        if (__float_as_uint(m1.w) & HIPRZ_NODE_LEAF) continue;
        const uint32_t c = __float_as_uint(m1.z);
        const float4 l0 = w.nodes32[2 * size_t(c)], l1 = w.nodes32[2 * size_t(c) + 1], r0 = w.nodes32[2 * size_t(c) + 2], r1 = w.nodes32[2 * size_t(c) + 3];
        WBox bb{{l0.x, l0.z, l1.x}, {l0.y, l0.w, l1.y}};
        const WBox sb{{r0.x, r0.z, r1.x}, {r0.y, r0.w, r1.y}};
        for (int a = 0; a < 3; ++a) {
            if (bb.mn[a] > sb.mn[a]) bb.mn[a] = sb.mn[a];
            if (bb.mx[a] < sb.mx[a]) bb.mx[a] = sb.mx[a];
        }
        write(slot, bb, c, __float_as_uint(m1.w));
    }
    *w.n_pairs_out = (n_nodes - 1u) / 2u;
    *w.n_order_out = n_order;
}


// =======================================================================================
// Binned surface-area build on the device (HIPRZ_TREE_DEVICE_SAH): the host's SahBuilder (hiprz_host.cpp: 16 bins per axis, leaves of at most
// 8 triangles, a node step priced at 4 triangle tests) as kernels.  Top phase, level by level, for nodes of more than kSahSmall triangles: every
// triangle adds itself to its node's bins (a node's triangles are one contiguous run of positions, so a workgroup of 256 positions meets at
// most 9 such nodes and keeps their bins in LDS before it touches the global ones), one thread per node evaluates the 3 x 15 planes and
// allocates the children, every triangle moves to its side (one atomic per wave, node and side).  Bottom phase: one thread per subtree of at
// most kSahSmall triangles runs the host's recursion on its own run of positions.  Boxes are then fitted exactly, bottom-up, by the refit kernel.
// =======================================================================================
constexpr uint32_t kSahBins = 16u, kSahLeaf = 8u, kSahSmall = 32u;
constexpr float kSahTraversal = 4.0f;
constexpr uint32_t kSahLarge = 1u, kSahSmallRoot = 2u, kSahLeafNode = 3u, kSahInner = 4u;  // node states (n_info & 7); axis in bits 4..5, plane in bits 8..12

struct SahViews {
    const float4* tris;
    const float4* attrs;
    uint32_t n;
    float* cen;             // [n][3] centre of the triangle's box
    float* tbox;            // [n][6] min.xyz max.xyz
    uint32_t* idx[2];       // triangle of the mesh at every position, ping-pong per level
    uint32_t* tri_node[2];  // node that holds the position
    uint32_t* n_first;      // nodes (local ids: 0 = root; a pair of children is allocated together)
    uint32_t* n_count;
    float* n_box;           // [6] the box the node bins over (exact boxes are fitted at the end)
    uint32_t* n_parent;
    uint32_t* n_info;
    uint32_t* n_child;
    uint32_t* n_hist;       // slot of the node in this level's bins
    uint32_t* hist;         // [slot][3][16][7]: count, ordered(min.xyz), ordered(max.xyz)
    uint32_t* cursors;      // [node][2]
    uint32_t* counters;     // [0] nodes allocated, [1] large nodes of the next level, [2] small roots
    uint32_t* active[2];
    uint32_t* small_roots;
    uint32_t leaf_max;      // kSahLeaf (experiments: HIPRZ_SAH_LEAF, 1..8)
    float traversal;        // kSahTraversal (experiments: HIPRZ_SAH_COST)
};
RZ_DEV uint32_t ordered(float f) {
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
RZ_DEV float unordered(uint32_t k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k); }
RZ_DEV float box_area(const float* b) {
    const float dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2];
    return dx * dy + dy * dz + dz * dx;
}
RZ_DEV void box_grow(float* b, const float* o) {
    for (int k = 0; k < 3; ++k) b[k] = fminf(b[k], o[k]), b[3 + k] = fmaxf(b[3 + k], o[3 + k]);
}
RZ_DEV int sah_bin(float c, float lo, float scale) {
    int b = int((c - lo) * scale);
    return b < 0 ? 0 : (b >= int(kSahBins) ? int(kSahBins) - 1 : b);
}

__global__ void __launch_bounds__(256) rz_sah_prepare_kernel(SahViews v, float3 mesh_lo, float3 mesh_hi) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t == 0u) {
        v.n_first[0] = 0u, v.n_count[0] = v.n, v.n_parent[0] = RZ_END, v.n_hist[0] = 0u, v.n_child[0] = 0u;
        v.n_info[0] = v.n > kSahSmall ? kSahLarge : kSahSmallRoot;
        const float b[6] = {mesh_lo.x, mesh_lo.y, mesh_lo.z, mesh_hi.x, mesh_hi.y, mesh_hi.z};
        for (int k = 0; k < 6; ++k) v.n_box[k] = b[k];
        v.counters[0] = 1u, v.counters[1] = 0u, v.counters[2] = 0u;
        v.active[0][0] = 0u;
        if (v.n <= kSahSmall) v.small_roots[0] = 0u, v.counters[2] = 1u;
    }
    if (t >= v.n) return;
    float mn[3], mx[3];
    triangle_box(v.tris, v.attrs, t, mn, mx);
    for (int k = 0; k < 3; ++k) v.tbox[6 * size_t(t) + k] = mn[k], v.tbox[6 * size_t(t) + 3 + k] = mx[k], v.cen[3 * size_t(t) + k] = (mn[k] + mx[k]) * 0.5f;
    v.idx[0][t] = t, v.tri_node[0][t] = 0u;
}
__global__ void __launch_bounds__(256) rz_sah_clear_kernel(uint32_t* hist, uint32_t n_slots) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n_slots * 3u * kSahBins * 7u) return;
    const uint32_t w = i % 7u;
    hist[i] = w == 0u ? 0u : (w < 4u ? 0xFFFFFFFFu : 0u);
}
__global__ void __launch_bounds__(256) rz_sah_bin_kernel(SahViews v, uint32_t cur) {
    __shared__ uint32_t lds_hist[9][3 * kSahBins][7];  // 256 positions meet at most 9 runs of more than 32 (two cut ones around seven whole ones)
    __shared__ uint32_t lds_node[9];
    __shared__ uint32_t wave_starts[4];
    const uint32_t p = blockIdx.x * 256u + threadIdx.x, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint32_t node = 0u;
    bool large = false;
    if (p < v.n) {
        node = v.tri_node[cur][p];
        large = (v.n_info[node] & 7u) == kSahLarge;
    }
    // a node's positions are one run, and only nodes of more than kSahSmall triangles take part: at most 8 runs per 256 positions
    const bool starts = large && (p == v.n_first[node] || tid == 0u);
    const unsigned long long smask = __ballot(starts);
    if (lane == 0u) wave_starts[wave] = uint32_t(__popcll(smask));
    for (uint32_t k = tid; k < 9u * 3u * kSahBins * 7u; k += 256u) {
        const uint32_t w = k % 7u;
        (&lds_hist[0][0][0])[k] = w == 0u ? 0u : (w < 4u ? 0xFFFFFFFFu : 0u);
    }
    __syncthreads();
    uint32_t local = uint32_t(__popcll(smask & ((2ull << lane) - 1ull)));
    for (uint32_t w = 0; w < wave; ++w) local += wave_starts[w];
    local -= 1u;  // runs before and including mine, minus one (a large position always has a start at or before it)
    if (starts) lds_node[local] = node;
    if (large) {
        const uint32_t t = v.idx[cur][p];
        const float* nb = v.n_box + 6 * size_t(node);
        const float* tb = v.tbox + 6 * size_t(t);
        for (uint32_t a = 0; a < 3u; ++a) {
            const float lo = nb[a], ext = nb[3 + a] - nb[a];
            const int b = sah_bin(v.cen[3 * size_t(t) + a], lo, ext > 0.0f ? float(kSahBins) / ext : 0.0f);
            uint32_t* h = lds_hist[local][a * kSahBins + uint32_t(b)];
            atomicAdd(&h[0], 1u);
            for (int k = 0; k < 3; ++k) atomicMin(&h[1 + k], ordered(tb[k])), atomicMax(&h[4 + k], ordered(tb[3 + k]));
        }
    }
    __syncthreads();
    const uint32_t n_local = wave_starts[0] + wave_starts[1] + wave_starts[2] + wave_starts[3];
    for (uint32_t k = tid; k < n_local * 3u * kSahBins; k += 256u) {
        const uint32_t l = k / (3u * kSahBins), ab = k % (3u * kSahBins);
        const uint32_t* h = lds_hist[l][ab];
        if (h[0] == 0u) continue;
        uint32_t* g = v.hist + (size_t(v.n_hist[lds_node[l]]) * 3u * kSahBins + ab) * 7u;
        atomicAdd(&g[0], h[0]);
        for (int q = 1; q < 4; ++q) atomicMin(&g[q], h[q]);
        for (int q = 4; q < 7; ++q) atomicMax(&g[q], h[q]);
    }
}
// one thread per large node of the level: the plane of least cost over the three axes (SahBuilder::build), or — no plane separates
// anything — the run cut in half; allocates the two children
__global__ void __launch_bounds__(64) rz_sah_split_kernel(SahViews v, uint32_t cur_list, uint32_t n_active) {
    const uint32_t a_i = blockIdx.x * 64u + threadIdx.x;
    if (a_i >= n_active) return;
    const uint32_t node = v.active[cur_list][a_i];
    const uint32_t* hist = v.hist + size_t(v.n_hist[node]) * 3u * kSahBins * 7u;
    const uint32_t count = v.n_count[node], first = v.n_first[node];
    float best_cost = 3.4e38f;
    int best_axis = -1, best_plane = 0;
    uint32_t best_left = 0u;
    float best_lbox[6] = {0, 0, 0, 0, 0, 0}, best_rbox[6] = {0, 0, 0, 0, 0, 0};
    for (int a = 0; a < 3; ++a) {
        const uint32_t* h = hist + size_t(a) * kSahBins * 7u;
        float right_area[kSahBins], acc[6];
        uint32_t right_count[kSahBins];
        uint32_t n = 0u;
        for (int b = int(kSahBins) - 1; b > 0; --b) {
            const uint32_t* hb = h + size_t(b) * 7u;
            if (hb[0]) {
                const float bb[6] = {unordered(hb[1]), unordered(hb[2]), unordered(hb[3]), unordered(hb[4]), unordered(hb[5]), unordered(hb[6])};
                if (n == 0u) for (int k = 0; k < 6; ++k) acc[k] = bb[k];
                else box_grow(acc, bb);
                n += hb[0];
            }
            right_area[b] = n ? box_area(acc) : 0.0f, right_count[b] = n;
        }
        n = 0u;
        for (int b = 0; b + 1 < int(kSahBins); ++b) {
            const uint32_t* hb = h + size_t(b) * 7u;
            if (hb[0]) {
                const float bb[6] = {unordered(hb[1]), unordered(hb[2]), unordered(hb[3]), unordered(hb[4]), unordered(hb[5]), unordered(hb[6])};
                if (n == 0u) for (int k = 0; k < 6; ++k) acc[k] = bb[k];
                else box_grow(acc, bb);
                n += hb[0];
            }
            if (n == 0u || right_count[b + 1] == 0u) continue;
            const float cost = box_area(acc) * float(n) + right_area[b + 1] * float(right_count[b + 1]);
            if (cost < best_cost) {
                best_cost = cost, best_axis = a, best_plane = b + 1, best_left = n;
                for (int k = 0; k < 6; ++k) best_lbox[k] = acc[k];
            }
        }
    }
    const float* nb = v.n_box + 6 * size_t(node);
    uint32_t axis = 3u, plane = 0u, left = count / 2u;  // no plane separates anything: halve the run (partition type "by size": never flipped)
    for (int k = 0; k < 6; ++k) best_rbox[k] = nb[k];
    if (best_axis >= 0) {
        axis = uint32_t(best_axis), plane = uint32_t(best_plane), left = best_left;
        const uint32_t* h = hist + size_t(best_axis) * kSahBins * 7u;
        bool any = false;
        for (uint32_t b = plane; b < kSahBins; ++b) {
            const uint32_t* hb = h + size_t(b) * 7u;
            if (!hb[0]) continue;
            const float bb[6] = {unordered(hb[1]), unordered(hb[2]), unordered(hb[3]), unordered(hb[4]), unordered(hb[5]), unordered(hb[6])};
            if (!any) for (int k = 0; k < 6; ++k) best_rbox[k] = bb[k];
            else box_grow(best_rbox, bb);
            any = true;
        }
    } else {
        for (int k = 0; k < 6; ++k) best_lbox[k] = nb[k];
    }
    const uint32_t pair = atomicAdd(&v.counters[0], 2u);
    v.n_child[node] = pair;
    v.n_info[node] = kSahInner | (axis << 4) | (plane << 8);
    v.cursors[2 * size_t(node)] = 0u, v.cursors[2 * size_t(node) + 1] = 0u;
    for (uint32_t k = 0; k < 2u; ++k) {
        const uint32_t c = pair + k, c_count = k == 0u ? left : count - left;
        v.n_first[c] = k == 0u ? first : first + left, v.n_count[c] = c_count, v.n_parent[c] = node, v.n_child[c] = 0u;
        for (int q = 0; q < 6; ++q) v.n_box[6 * size_t(c) + q] = k == 0u ? best_lbox[q] : best_rbox[q];
        if (c_count > kSahSmall) {
            const uint32_t slot = atomicAdd(&v.counters[1], 1u);
            v.active[cur_list ^ 1u][slot] = c, v.n_hist[c] = slot, v.n_info[c] = kSahLarge;
        } else {
            v.small_roots[atomicAdd(&v.counters[2], 1u)] = c, v.n_info[c] = kSahSmallRoot;
        }
    }
}
// every position moves to its side of its node's plane (positions of nodes that were not split this level stay)
__global__ void __launch_bounds__(256) rz_sah_partition_kernel(SahViews v, uint32_t cur) {
    const uint32_t p = blockIdx.x * 256u + threadIdx.x, lane = threadIdx.x & 63u;
    const bool in = p < v.n;
    uint32_t node = in ? v.tri_node[cur][p] : RZ_END, t = in ? v.idx[cur][p] : 0u;
    const uint32_t info = in ? v.n_info[node] : 0u;
    const bool moved = in && (info & 7u) == kSahInner;  // a triangle points at an inner node only in the level that split it
    uint32_t side = 0u, dest = p, child = node;
    if (moved) {
        const uint32_t axis = (info >> 4) & 3u, plane = (info >> 8) & 31u, first = v.n_first[node], count = v.n_count[node];
        if (axis == 3u) side = (p - first) >= count / 2u ? 1u : 0u;
        else {
            const float* nb = v.n_box + 6 * size_t(node);
            const float ext = nb[3 + axis] - nb[axis];
            side = uint32_t(sah_bin(v.cen[3 * size_t(t) + axis], nb[axis], ext > 0.0f ? float(kSahBins) / ext : 0.0f)) >= plane ? 1u : 0u;
        }
        child = v.n_child[node] + side;
    }
    // one atomic per (wave, node, side): the lanes of a group count themselves by ballot
    unsigned long long todo = __ballot(moved);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t key = __shfl(child, leader);
        const unsigned long long group = __ballot(moved && child == key);
        if (moved && child == key) {
            uint32_t base = 0u;
            if (int(lane) == leader) base = atomicAdd(&v.cursors[2 * size_t(node) + side], uint32_t(__popcll(group)));
            base = __shfl(base, leader);
            dest = v.n_first[child] + base + uint32_t(__popcll(group & ((1ull << lane) - 1ull)));
        }
        todo &= ~group;
    }
    if (in) v.idx[cur ^ 1u][dest] = t, v.tri_node[cur ^ 1u][dest] = child;
}
// one thread per subtree of at most kSahSmall triangles: SahBuilder::build on its own run of positions
__global__ void __launch_bounds__(64) rz_sah_small_kernel(SahViews v, uint32_t cur, uint32_t n_roots) {
    const uint32_t r = blockIdx.x * 64u + threadIdx.x;
    if (r >= n_roots) return;
    uint32_t* idx = v.idx[cur];
    uint32_t stack[2 * kSahSmall];
    uint32_t sp = 0u;
    stack[sp++] = v.small_roots[r];
    {   // The top phase places triangles with per-wave atomic cursors: WHICH triangles a small root holds is decided by planes (deterministic),
        // the ORDER inside its run by wave scheduling.  The bottom phase reads that order twice (the two-pointer partition fixes the leaf
        // order; coincident centroids are cut "as the run stands"), so the run is put into triangle order first: the subtree is then a function
        // of its triangle set, the same from run to run and on every device of a context.  (What stays order-dependent: a LARGE node whose
        // centroids no plane separates is cut in half as its run stands — rz_sah_split_kernel; meshes with > 32 coincident centroids only.)
        const uint32_t first = v.n_first[stack[0]], count = v.n_count[stack[0]];
        for (uint32_t i = 1u; i < count; ++i) {
            const uint32_t t = idx[first + i];
            uint32_t j = i;
            for (; j > 0u && idx[first + j - 1u] > t; --j) idx[first + j] = idx[first + j - 1u];
            idx[first + j] = t;
        }
    }
    while (sp) {
        const uint32_t node = stack[--sp], first = v.n_first[node], count = v.n_count[node];
        float cbox[6];
        for (uint32_t k = 0; k < count; ++k) {
            const float* c = v.cen + 3 * size_t(idx[first + k]);
            for (int q = 0; q < 3; ++q) cbox[q] = k ? fminf(cbox[q], c[q]) : c[q], cbox[3 + q] = k ? fmaxf(cbox[3 + q], c[q]) : c[q];
        }
        float box[6];
        for (uint32_t k = 0; k < count; ++k) {
            const float* tb = v.tbox + 6 * size_t(idx[first + k]);
            if (k == 0u) for (int q = 0; q < 6; ++q) box[q] = tb[q];
            else box_grow(box, tb);
        }
        for (int q = 0; q < 6; ++q) v.n_box[6 * size_t(node) + q] = box[q];
        bool leaf = count <= 2u;
        int best_axis = -1, best_plane = 0;
        float best_cost = 3.4e38f;
        if (!leaf) {
            for (int a = 0; a < 3; ++a) {
                const float lo = cbox[a], ext = cbox[3 + a] - cbox[a];
                if (!(ext > 0.0f)) continue;
                const float scale = float(kSahBins) / ext;
                float bin_box[kSahBins][6];
                uint32_t bin_count[kSahBins];
                for (uint32_t b = 0; b < kSahBins; ++b) bin_count[b] = 0u;
                for (uint32_t k = 0; k < count; ++k) {
                    const uint32_t t = idx[first + k];
                    const int b = sah_bin(v.cen[3 * size_t(t) + a], lo, scale);
                    const float* tb = v.tbox + 6 * size_t(t);
                    if (bin_count[b]++ == 0u) for (int q = 0; q < 6; ++q) bin_box[b][q] = tb[q];
                    else box_grow(bin_box[b], tb);
                }
                float right_area[kSahBins], acc[6];
                uint32_t right_count[kSahBins], n = 0u;
                for (int b = int(kSahBins) - 1; b > 0; --b) {
                    if (bin_count[b]) {
                        if (n == 0u) for (int q = 0; q < 6; ++q) acc[q] = bin_box[b][q];
                        else box_grow(acc, bin_box[b]);
                        n += bin_count[b];
                    }
                    right_area[b] = n ? box_area(acc) : 0.0f, right_count[b] = n;
                }
                n = 0u;
                for (int b = 0; b + 1 < int(kSahBins); ++b) {
                    if (bin_count[b]) {
                        if (n == 0u) for (int q = 0; q < 6; ++q) acc[q] = bin_box[b][q];
                        else box_grow(acc, bin_box[b]);
                        n += bin_count[b];
                    }
                    if (n == 0u || right_count[b + 1] == 0u) continue;
                    const float cost = box_area(acc) * float(n) + right_area[b + 1] * float(right_count[b + 1]);
                    if (cost < best_cost) best_cost = cost, best_axis = a, best_plane = b + 1;
                }
            }
            const float node_area = box_area(box);
            const float split_cost = best_axis >= 0 && node_area > 0.0f ? v.traversal + best_cost / node_area : 3.4e38f;
            if (count <= v.leaf_max && float(count) <= split_cost) leaf = true;
            if (!leaf && !(best_axis >= 0 && (count > v.leaf_max || split_cost < float(count)))) best_axis = -1;
        }
        if (leaf) {
            v.n_info[node] = kSahLeafNode;
            continue;
        }
        uint32_t mid, axis = 3u;
        if (best_axis >= 0) {  // two-pointer partition of the run by bin < plane
            axis = uint32_t(best_axis);
            const float lo = cbox[best_axis], scale = float(kSahBins) / (cbox[3 + best_axis] - cbox[best_axis]);
            uint32_t i = first, j = first + count;
            while (i < j) {
                if (sah_bin(v.cen[3 * size_t(idx[i]) + best_axis], lo, scale) < best_plane) ++i;
                else {
                    --j;
                    const uint32_t tmp = idx[i];
                    idx[i] = idx[j], idx[j] = tmp;
                }
            }
            mid = i - first;
        } else {  // every centroid in one spot: halve the run as it stands
            mid = count / 2u;
        }
        if (mid == 0u || mid == count) {
            v.n_info[node] = kSahLeafNode;  // (cannot happen: a chosen plane has triangles on both sides)
            continue;
        }
        const uint32_t pair = atomicAdd(&v.counters[0], 2u);
        v.n_child[node] = pair, v.n_info[node] = kSahInner | (axis << 4);
        for (uint32_t k = 0; k < 2u; ++k) {
            const uint32_t c = pair + k;
            v.n_first[c] = k == 0u ? first : first + mid, v.n_count[c] = k == 0u ? mid : count - mid, v.n_parent[c] = node, v.n_child[c] = 0u;
            v.n_info[c] = kSahSmallRoot;
        }
        stack[sp++] = pair + 1u, stack[sp++] = pair;
    }
}
// the node records of the region in the walks' layout (boxes provisional: the refit kernel fits them exactly afterwards)
__global__ void __launch_bounds__(256) rz_sah_emit_kernel(SahViews v, EmitViews e) {
    const uint32_t k = blockIdx.x * 256u + threadIdx.x, n_nodes = v.counters[0];
    if (k == 0u) v.counters[3] = (n_nodes - 1u) / 2u;  // pairs of children, as the links kernel counts
    if (k >= n_nodes) return;
    const uint32_t info = v.n_info[k], state = info & 7u;
    if (state == kSahInner) {
        const uint32_t axis = (info >> 4) & 3u;
        write_node(e, e.region + k, v.n_box + 6 * size_t(k), e.region + v.n_child[k], (axis == 3u ? 3u : 2u - axis) << HIPRZ_NODE_PTYPE_SHIFT);
    } else {
        write_node(e, e.region + k, v.n_box + 6 * size_t(k), e.tri_first + v.n_first[k], v.n_count[k] | HIPRZ_NODE_LEAF);
    }
    e.slot_parent[k] = v.n_parent[k] == RZ_END ? RZ_END : e.region + v.n_parent[k];
}

// Host proof over the downloaded tables of one region (what derive_tables proves for host-built trees): under every octant the walk
// that enters every box visits each of the region's nodes exactly once and ends; every index stays inside the region; the leaves
// tile the mesh's triangle range exactly.
bool validate_region(const std::vector<uint32_t>& rec64, uint32_t region, uint32_t n_slots, uint32_t tri_first, uint32_t n_tris, uint32_t max_leaf, std::string& why) {
    if (std::getenv("HIPRZ_TEST_REFUSE_TREES")) return why = "refused on request (HIPRZ_TEST_REFUSE_TREES: the tests' stand-in for a tree the proof rejects)", false;
    auto word = [&](uint32_t slot, uint32_t w) { return rec64[16 * size_t(slot - region) + w]; };
    auto inside = [&](uint32_t slot) { return slot >= region && slot < region + n_slots; };
    // one walk per ray octant, each on a thread of its own (the walks only read): 8 x n_slots dependent steps are what an upload with
    // device-built trees waits for (config D: 16 ms on one thread)
    const char* failed[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    std::vector<uint8_t> covered(n_tris, 0);  // written by octant 0's walk only
    auto walk = [&](uint32_t o) {
        std::vector<uint8_t> seen(n_slots, 0);
        uint32_t x = region, steps = 0;
        while (x != RZ_END) {
            if (!inside(x)) return void(failed[o] = "link leaves the mesh's region");
            if (seen[x - region]) return void(failed[o] = "a node is visited twice");
            seen[x - region] = 1;
            if (++steps > n_slots) return void(failed[o] = "walk does not end");
            const uint32_t begin = word(x, 6), meta = word(x, 7);
            if (meta & HIPRZ_NODE_LEAF) {
                const uint32_t count = meta & HIPRZ_NODE_COUNT_MASK;
                if (begin < tri_first || uint64_t(begin) + count > uint64_t(tri_first) + n_tris || count == 0u || count > max_leaf) return void(failed[o] = "leaf range outside the mesh");
                if (o == 0u)
                    for (uint32_t t = begin; t < begin + count; ++t) {
                        if (covered[t - tri_first]) return void(failed[o] = "two leaves share a triangle");
                        covered[t - tri_first] = 1;
                    }
                x = word(x, 8u + o);
            } else {
                const uint32_t ptype = (meta >> HIPRZ_NODE_PTYPE_SHIFT) & 3u;
                if (!inside(begin) || !inside(begin + 1u)) return void(failed[o] = "children outside the region");
                x = begin + ((o >> ptype) & 1u);
            }
        }
        if (steps != n_slots) failed[o] = "a walk misses nodes";
    };
    if (n_slots < 4096u) {
        for (uint32_t o = 0; o < 8u; ++o) walk(o);
    } else {
        std::thread threads[7];
        for (uint32_t o = 1; o < 8u; ++o) threads[o - 1u] = std::thread(walk, o);
        walk(0u);
        for (auto& t : threads) t.join();
    }
    for (uint32_t o = 0; o < 8u; ++o)
        if (failed[o]) return why = failed[o], false;
    for (uint32_t t = 0; t < n_tris; ++t)
        if (!covered[t]) return why = "a triangle is in no leaf", false;
    return true;
}

template <typename T>
T* carve(unsigned char*& cursor, size_t count) {
    T* p = reinterpret_cast<T*>(cursor);
    cursor += ((count * sizeof(T) + 255u) / 256u) * 256u;
    return p;
}

// leaves of at most this many triangles (experiments: HIPRZ_SAH_LEAF, 1 .. kSahSmall; the walks take a leaf's triangles 8 per round)
uint32_t sah_leaf_max() {
    if (const char* e = std::getenv("HIPRZ_SAH_LEAF")) return std::min(kSahSmall, std::max(1u, uint32_t(std::atoi(e))));
    return kSahLeaf;
}
constexpr size_t sah_workspace_bytes(size_t n) {
    return 256u * 40u + n * 4u * (3u + 6u + 2u + 2u + 1u) + 2u * n * 4u * (6u + 6u + 2u) + (n / kSahSmall + 2u) * (3u * kSahBins * 7u * 4u + 8u) + n * 16u * 9u;
}
// one mesh by the binned surface-area build; the stream is synchronised once per level of the top phase (the host sizes the next level)
int build_mesh_sah(hiprz_ctx* c, DeviceMesh& m, float4* blob_tris, float4* blob_attrs, size_t n_cap) {
    hipStream_t st = c->stream;
    unsigned char* cursor = c->build_temp.ptr;
    const size_t n = n_cap, nodes = 2u * n_cap, slots = n_cap / kSahSmall + 2u;
    SahViews v{};
    v.tris = blob_tris + 3 * size_t(m.tri_first), v.attrs = blob_attrs + 6 * size_t(m.tri_first), v.n = m.n_tris;
    v.cen = carve<float>(cursor, 3 * n), v.tbox = carve<float>(cursor, 6 * n);
    v.idx[0] = carve<uint32_t>(cursor, n), v.idx[1] = carve<uint32_t>(cursor, n);
    v.tri_node[0] = carve<uint32_t>(cursor, n), v.tri_node[1] = carve<uint32_t>(cursor, n);
    v.n_first = carve<uint32_t>(cursor, nodes), v.n_count = carve<uint32_t>(cursor, nodes), v.n_parent = carve<uint32_t>(cursor, nodes);
    v.n_info = carve<uint32_t>(cursor, nodes), v.n_child = carve<uint32_t>(cursor, nodes), v.n_hist = carve<uint32_t>(cursor, nodes);
    v.n_box = carve<float>(cursor, 6 * nodes), v.cursors = carve<uint32_t>(cursor, 2 * nodes);
    v.hist = carve<uint32_t>(cursor, slots * 3u * kSahBins * 7u);
    v.active[0] = carve<uint32_t>(cursor, slots), v.active[1] = carve<uint32_t>(cursor, slots);
    v.small_roots = carve<uint32_t>(cursor, n);
    v.counters = carve<uint32_t>(cursor, 8);
    float4* tris_tmp = carve<float4>(cursor, 3 * n);
    float4* attrs_tmp = carve<float4>(cursor, 6 * n);
    v.leaf_max = sah_leaf_max(), v.traversal = kSahTraversal;
    if (const char* e = std::getenv("HIPRZ_SAH_COST")) v.traversal = float(std::atof(e));
    const uint32_t blocks = (m.n_tris + 255u) / 256u;
    RZ_LAUNCH(rz_sah_prepare_kernel, dim3(blocks), dim3(256), 0, st, v, make_float3(m.bb_min[0], m.bb_min[1], m.bb_min[2]),
                       make_float3(m.bb_max[0], m.bb_max[1], m.bb_max[2]));
    uint32_t cur = 0u, list = 0u, n_active = m.n_tris > kSahSmall ? 1u : 0u;
    for (uint32_t level = 0; n_active != 0u; ++level) {
        if (level > 256u) return fail(c, HIPRZ_ERR_DEVICE, "device surface-area build does not end");
        RZ_LAUNCH(rz_sah_clear_kernel, dim3((n_active * 3u * kSahBins * 7u + 255u) / 256u), dim3(256), 0, st, v.hist, n_active);
        RZ_LAUNCH(rz_sah_bin_kernel, dim3(blocks), dim3(256), 0, st, v, cur);
        RZ_LAUNCH(rz_sah_split_kernel, dim3((n_active + 63u) / 64u), dim3(64), 0, st, v, list, n_active);
        RZ_LAUNCH(rz_sah_partition_kernel, dim3(blocks), dim3(256), 0, st, v, cur);
        RZ_HIP(c, hipMemcpyAsync(&n_active, v.counters + 1, 4, hipMemcpyDeviceToHost, st));
        RZ_HIP(c, hipMemsetAsync(v.counters + 1, 0, 4, st));
        RZ_HIP(c, hipStreamSynchronize(st));
        if (n_active > slots) return fail(c, HIPRZ_ERR_DEVICE, "device surface-area build: more large nodes than triangles allow");
        cur ^= 1u, list ^= 1u;
    }
    uint32_t n_roots = 0u;
    RZ_HIP(c, hipMemcpyAsync(&n_roots, v.counters + 2, 4, hipMemcpyDeviceToHost, st));
    RZ_HIP(c, hipStreamSynchronize(st));
    if (n_roots > m.n_tris) return fail(c, HIPRZ_ERR_DEVICE, "device surface-area build: more subtrees than triangles");
    if (n_roots) RZ_LAUNCH(rz_sah_small_kernel, dim3((n_roots + 63u) / 64u), dim3(64), 0, st, v, cur, n_roots);
    EmitViews e{reinterpret_cast<float4*>(c->dev_nodes.ptr), c->slot_parent.ptr + m.region, m.region, m.tri_first};
    const uint32_t max_slots = 2u * m.n_tris - 1u;
    RZ_LAUNCH(rz_sah_emit_kernel, dim3((max_slots + 255u) / 256u), dim3(256), 0, st, v, e);
    BuildViews b{};
    b.tris = v.tris, b.attrs = v.attrs, b.n = m.n_tris, b.perm = v.idx[cur];
    RZ_LAUNCH(rz_build_permute_kernel, dim3(blocks), dim3(256), 0, st, b, tris_tmp, attrs_tmp);
    RZ_HIP(c, hipMemcpyAsync(const_cast<float4*>(v.tris), tris_tmp, size_t(m.n_tris) * 48u, hipMemcpyDeviceToDevice, st));
    RZ_HIP(c, hipMemcpyAsync(const_cast<float4*>(v.attrs), attrs_tmp, size_t(m.n_tris) * 96u, hipMemcpyDeviceToDevice, st));
    uint32_t n_nodes = 0u;
    RZ_HIP(c, hipMemcpyAsync(&n_nodes, v.counters, 4, hipMemcpyDeviceToHost, st));
    RZ_HIP(c, hipStreamSynchronize(st));
    if (n_nodes == 0u || n_nodes > max_slots || !(n_nodes & 1u)) return fail(c, HIPRZ_ERR_DEVICE, "device surface-area build: node count out of range");
    m.n_slots = n_nodes;
    // exact boxes bottom-up over the emitted topology, then the skip links and the 64-byte walk records
    RZ_HIP(c, c->refit_visit.resize(m.n_slots));
    RZ_HIP(c, hipMemsetAsync(c->refit_visit.ptr, 0, size_t(m.n_slots) * 4u, st));
    RZ_LAUNCH(rz_refit_kernel, dim3((m.n_slots + 255u) / 256u), dim3(256), 0, st, e, m.n_slots, blob_tris, blob_attrs, c->refit_visit.ptr, c->nodes64.ptr);
    RZ_LAUNCH(rz_build_links_kernel, dim3((m.n_slots * 8u + 255u) / 256u), dim3(256), 0, st, e, v.counters + 3, c->nodes64.ptr, c->node_skip.ptr);
    RZ_HIP(c, hipStreamSynchronize(st));
    RZ_HIP(c, hipGetLastError());
    return HIPRZ_OK;
}

}  // namespace

// Capacity of the node arrays of a scene whose mesh trees are built on the device: the uploaded prefix + per mesh a region of 2n - 1
// slots starting at an odd index (its child pairs then start at even indices: one 128-byte line per pair of 64-byte walk records).
uint32_t device_build_regions(std::vector<DeviceMesh>& meshes, uint32_t prefix_nodes) {
    uint32_t cursor = prefix_nodes;
    for (auto& m : meshes) {
        if (m.n_tris <= kLeafMax) {
            m.region = RZ_END;  // stays the single leaf of the uploaded placeholder
            continue;
        }
        if (!(cursor & 1u)) cursor += 1u;
        m.region = cursor;
        cursor += 2u * m.n_tris - 1u;
    }
    return cursor;
}

// Builds the trees of `meshes` on the device into the scene's node arrays (c->dev_nodes, c->node_skip, c->nodes64, sized by the caller
// through device_build_regions), reorders the triangles of the hot blob accordingly and re-points the instances.  Synchronous.
int device_build_mesh_trees(hiprz_ctx* c, std::vector<DeviceMesh>& meshes, const std::vector<uint32_t>& instance_mesh, bool validate) {
    StageTimer timer;
    hipStream_t st = c->stream;
    float4* blob_tris = reinterpret_cast<float4*>(c->hot.ptr + c->dscene.off_tris);
    float4* blob_attrs = reinterpret_cast<float4*>(c->hot.ptr + c->dscene.off_tri_attrs);
    uint32_t n_max = 0;
    for (const auto& m : meshes) n_max = std::max(n_max, m.n_tris);
    // one workspace, carved; every array padded to 256 bytes (meshes of at most kLeafMax triangles stay the single leaves they were uploaded as)
    const size_t n = std::max<uint32_t>(n_max, kLeafMax + 1u);
    const size_t bytes = c->build_sah ? sah_workspace_bytes(n) : 256u * 24u + n * 4u * 12u + n * 4u * 6u * 2u + n * 16u * 9u + (2u * n) * 4u;
    RZ_HIP(c, c->build_temp.resize(bytes));
    RZ_HIP(c, c->slot_parent.resize(c->node_capacity));
    const int src = sort_temp_resize(c, c->build_sort, n);
    if (src != HIPRZ_OK) return src;
    for (auto& m : meshes) {
        if (m.region == RZ_END) continue;
        if (c->build_sah) {
            const int rc = build_mesh_sah(c, m, blob_tris, blob_attrs, n);
            if (rc != HIPRZ_OK) return rc;
            if (validate) {
                std::vector<uint32_t> rec(16 * size_t(m.n_slots));
                RZ_HIP(c, hipMemcpy(rec.data(), c->nodes64.ptr + 16 * size_t(m.region), rec.size() * 4u, hipMemcpyDeviceToHost));
                std::string why;
                if (!validate_region(rec, m.region, m.n_slots, m.tri_first, m.n_tris, sah_leaf_max(), why))
                    return fail(c, HIPRZ_ERR_DEVICE, "device-built mesh tree refused (" + why + ")");
            }
            continue;
        }
        unsigned char* cursor = c->build_temp.ptr;
        BuildViews b{};
        b.tris = blob_tris + 3 * size_t(m.tri_first), b.attrs = blob_attrs + 6 * size_t(m.tri_first), b.n = m.n_tris;
        b.keys = carve<uint32_t>(cursor, n), b.sorted_keys = carve<uint32_t>(cursor, n), b.perm = carve<uint32_t>(cursor, n);
        b.left = carve<uint32_t>(cursor, n), b.right = carve<uint32_t>(cursor, n), b.parent = carve<uint32_t>(cursor, n);
        b.leaf_parent = carve<uint32_t>(cursor, n), b.first = carve<uint32_t>(cursor, n), b.count = carve<uint32_t>(cursor, n);
        b.visit = carve<uint32_t>(cursor, n), b.survive = carve<uint32_t>(cursor, n), b.rank = carve<uint32_t>(cursor, n + 1), b.info = carve<uint32_t>(cursor, n);
        b.ibox = carve<float>(cursor, 6 * n), b.lbox = carve<float>(cursor, 6 * n);
        float4* tris_tmp = carve<float4>(cursor, 3 * n);
        float4* attrs_tmp = carve<float4>(cursor, 6 * n);
        const uint32_t blocks = (m.n_tris + 255u) / 256u;
        float3 lo = make_float3(m.bb_min[0], m.bb_min[1], m.bb_min[2]), scale;
        scale.x = m.bb_max[0] > m.bb_min[0] ? 1024.0f / (m.bb_max[0] - m.bb_min[0]) : 0.0f;
        scale.y = m.bb_max[1] > m.bb_min[1] ? 1024.0f / (m.bb_max[1] - m.bb_min[1]) : 0.0f;
        scale.z = m.bb_max[2] > m.bb_min[2] ? 1024.0f / (m.bb_max[2] - m.bb_min[2]) : 0.0f;
        RZ_LAUNCH(rz_build_morton_kernel, dim3(blocks), dim3(256), 0, st, b, lo, scale);
        sort_u32(st, b.keys, m.n_tris, 32, b.perm, b.sorted_keys, c->build_sort);
        RZ_LAUNCH(rz_build_radix_tree_kernel, dim3(blocks), dim3(256), 0, st, b);
        RZ_LAUNCH(rz_build_fit_kernel, dim3(blocks), dim3(256), 0, st, b);
        RZ_LAUNCH(rz_build_decide_kernel, dim3(blocks), dim3(256), 0, st, b);
        RZ_LAUNCH(rz_build_scan_kernel, dim3(1), dim3(1024), 0, st, b.survive, b.rank, m.n_tris - 1u);
        EmitViews e{reinterpret_cast<float4*>(c->dev_nodes.ptr), c->slot_parent.ptr + m.region, m.region, m.tri_first};
        RZ_LAUNCH(rz_build_emit_kernel, dim3(blocks), dim3(256), 0, st, b, e);
        const uint32_t max_slots = 2u * m.n_tris - 1u;
        RZ_LAUNCH(rz_build_links_kernel, dim3((max_slots * 8u + 255u) / 256u), dim3(256), 0, st, e, b.rank + (m.n_tris - 1u), c->nodes64.ptr, c->node_skip.ptr);
        RZ_LAUNCH(rz_build_permute_kernel, dim3(blocks), dim3(256), 0, st, b, tris_tmp, attrs_tmp);
        RZ_HIP(c, hipMemcpyAsync(const_cast<float4*>(b.tris), tris_tmp, size_t(m.n_tris) * 48u, hipMemcpyDeviceToDevice, st));
        RZ_HIP(c, hipMemcpyAsync(const_cast<float4*>(b.attrs), attrs_tmp, size_t(m.n_tris) * 96u, hipMemcpyDeviceToDevice, st));
        uint32_t surviving = 0u;
        RZ_HIP(c, hipMemcpyAsync(&surviving, b.rank + (m.n_tris - 1u), 4, hipMemcpyDeviceToHost, st));
        RZ_HIP(c, hipStreamSynchronize(st));
        RZ_HIP(c, hipGetLastError());
        m.n_slots = 1u + 2u * surviving;
        if (validate) {
            std::vector<uint32_t> rec(16 * size_t(m.n_slots));
            RZ_HIP(c, hipMemcpy(rec.data(), c->nodes64.ptr + 16 * size_t(m.region), rec.size() * 4u, hipMemcpyDeviceToHost));
            std::string why;
            if (!validate_region(rec, m.region, m.n_slots, m.tri_first, m.n_tris, kLeafMax, why))
                return fail(c, HIPRZ_ERR_DEVICE, "device-built mesh tree refused (" + why + ")");
        }
    }
    // instances enter their mesh at its new root
    for (size_t i = 0; i < instance_mesh.size(); ++i) {
        if (instance_mesh[i] == RZ_END || meshes[instance_mesh[i]].region == RZ_END) continue;
        const uint32_t root = meshes[instance_mesh[i]].region;
        RZ_HIP(c, hipMemcpyAsync(c->hot.ptr + c->dscene.off_instances + sizeof(hiprz_instance) * i + offsetof(hiprz_instance, blas_root), &root, 4, hipMemcpyHostToDevice, st));
    }
    RZ_HIP(c, c->ref_to_dev.resize(c->n_tris));
    if (c->n_tris) RZ_LAUNCH(rz_build_refpos_kernel, dim3((c->n_tris + 255u) / 256u), dim3(256), 0, st, blob_tris, c->n_tris, c->ref_to_dev.ptr);
    RZ_HIP(c, hipStreamSynchronize(st));
    c->device_meshes = meshes;
    c->timings.set("build mesh trees (device)", timer.ms());
    return HIPRZ_OK;
}

// The world tree of the instance records that are in the hot blob now, rebuilt by the device into its region of the node arrays
// (c->world_region, 2 * n_instances + 1 slots), its leaf order into the blob's tlas_order.
int device_build_world_tree(hiprz_ctx* c, bool validate) {
    StageTimer timer;
    hipStream_t st = c->stream;
    const uint32_t n = c->dscene.n_instances;
    if (n == 0u) return HIPRZ_OK;
    RZ_HIP(c, c->world_items.resize(size_t(n) + 2u));
    WorldViews w{};
    w.instances = reinterpret_cast<const float4*>(c->hot.ptr + c->dscene.off_instances), w.has_mesh = c->has_mesh.ptr, w.n_instances = n;
    w.items = c->world_items.ptr;
    w.order = reinterpret_cast<uint32_t*>(c->hot.ptr + c->dscene.off_tlas_order);
    w.nodes32 = reinterpret_cast<float4*>(c->dev_nodes.ptr), w.slot_parent = c->slot_parent.ptr + c->world_region, w.region = c->world_region;
    w.n_pairs_out = c->world_items.ptr + n, w.n_order_out = c->world_items.ptr + n + 1u;
    RZ_LAUNCH(rz_build_world_tree_kernel, dim3(1), dim3(64), 0, st, w);
    EmitViews e{reinterpret_cast<float4*>(c->dev_nodes.ptr), c->slot_parent.ptr + c->world_region, c->world_region, 0u};
    const uint32_t max_slots = 2u * n + 1u;
    RZ_LAUNCH(rz_build_links_kernel, dim3((max_slots * 8u + 255u) / 256u), dim3(256), 0, st, e, w.n_pairs_out, c->nodes64.ptr, c->node_skip.ptr);
    uint32_t out[2] = {0u, 0u};
    RZ_HIP(c, hipMemcpyAsync(out, w.n_pairs_out, 8, hipMemcpyDeviceToHost, st));
    RZ_HIP(c, hipStreamSynchronize(st));
    RZ_HIP(c, hipGetLastError());
    c->world_slots = 1u + 2u * out[0];
    if (out[1] != c->n_tlas_order) return fail(c, HIPRZ_ERR_DEVICE, "device-built world tree lists another number of instances than the scene");
    if (validate) {
        std::vector<uint32_t> rec(16 * size_t(c->world_slots));
        RZ_HIP(c, hipMemcpy(rec.data(), c->nodes64.ptr + 16 * size_t(c->world_region), rec.size() * 4u, hipMemcpyDeviceToHost));
        std::string why;
        if (!validate_region(rec, c->world_region, c->world_slots, 0u, c->n_tlas_order, 0x1FFFFFFFu, why))
            return fail(c, HIPRZ_ERR_DEVICE, "device-built world tree refused (" + why + ")");
    }
    c->timings.set("build world tree (device)", timer.ms());
    return HIPRZ_OK;
}

// hiprz_update_triangles: new vertices / shading records for triangles [first, first + n) of the uploaded snapshot's order, then the
// boxes of every device-built mesh tree that holds one of them are fitted again (topology and leaf assignment unchanged).
int device_update_triangles(hiprz_ctx* c, uint32_t first, uint32_t n, const hiprz_tri* tris, const hiprz_tri_attr* attrs) {
    StageTimer timer;
    hipStream_t st = c->stream;
    float4* blob_tris = reinterpret_cast<float4*>(c->hot.ptr + c->dscene.off_tris);
    float4* blob_attrs = reinterpret_cast<float4*>(c->hot.ptr + c->dscene.off_tri_attrs);
    RZ_HIP(c, c->update_tris.assign(tris, n, st));
    RZ_HIP(c, c->update_attrs.assign(attrs, n, st));
    RZ_LAUNCH(rz_refit_scatter_kernel, dim3((n + 255u) / 256u), dim3(256), 0, st, c->update_tris.ptr, c->update_attrs.ptr, first, n, c->ref_to_dev.ptr, blob_tris, blob_attrs);
    for (const auto& m : c->device_meshes) {
        if (m.n_tris == 0u || m.ref_first + m.n_tris <= first || first + n <= m.ref_first) continue;
        // a mesh too small to have been built is one leaf (the uploaded placeholder): a region of one slot without a parent
        const uint32_t region = m.region != RZ_END ? m.region : m.leaf_slot, n_slots = m.region != RZ_END ? m.n_slots : 1u;
        if (region == RZ_END) continue;
        RZ_HIP(c, c->refit_visit.resize(n_slots));
        RZ_HIP(c, hipMemsetAsync(c->refit_visit.ptr, 0, size_t(n_slots) * 4u, st));
        EmitViews e{reinterpret_cast<float4*>(c->dev_nodes.ptr), c->slot_parent.ptr + region, region, m.tri_first};
        RZ_LAUNCH(rz_refit_kernel, dim3((n_slots + 255u) / 256u), dim3(256), 0, st, e, n_slots, blob_tris, blob_attrs, c->refit_visit.ptr, c->nodes64.ptr);
    }
    RZ_HIP(c, hipStreamSynchronize(st));
    RZ_HIP(c, hipGetLastError());
    c->timings.set("refit mesh trees (device)", timer.ms());
    return HIPRZ_OK;
}

}  // namespace hiprz
