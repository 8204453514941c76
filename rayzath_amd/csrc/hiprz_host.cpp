// hiprz_host.cpp — host-side (pure CPU) half of libhiprz.so: tree builders that emit the
// flattened node layout of include/hiprz.h, instance bounds, axis construction and the
// seed table.  No device code here.
//
// The builders restate RayZath's host BVH construction — TreeNode::construct
// (RayZath/bvh_tree_node.hpp:117-215) and ComponentTreeNode::construct
// (RayZath/component_container.hpp:259-363) — but build straight into the flat array
// (no pointer tree): a node's two child slots are reserved when the node is opened, the
// first subtree is completed before the second, and boxes are fitted bottom-up on the way
// back, which is the order the reference's constructors run in.
#include "hiprz.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

struct Box {
    float mn[3], mx[3];
};

// BoundingBox(p1, p2): std::min / std::max per axis (RayZath/render_parts.cpp:166-177)
inline Box box_of(const float* p1, const float* p2) {
    Box b;
    for (int a = 0; a < 3; ++a) {
        b.mn[a] = std::min(p1[a], p2[a]);
        b.mx[a] = std::max(p1[a], p2[a]);
    }
    return b;
}
// BoundingBox::extendBy (RayZath/render_parts.cpp:181-197)
inline void grow(Box& b, const float* p) {
    for (int a = 0; a < 3; ++a) {
        if (b.mn[a] > p[a]) b.mn[a] = p[a];
        if (b.mx[a] < p[a]) b.mx[a] = p[a];
    }
}
inline void grow(Box& b, const Box& o) {
    for (int a = 0; a < 3; ++a) {
        if (b.mn[a] > o.mn[a]) b.mn[a] = o.mn[a];
        if (b.mx[a] < o.mx[a]) b.mx[a] = o.mx[a];
    }
}
inline float centroid(const Box& b, int a) { return (b.mn[a] + b.mx[a]) * 0.5f; }

class FlatTreeBuilder {
public:
    FlatTreeBuilder(const std::vector<Box>& boxes, uint32_t leaf_size, uint32_t root_leaf_size,
                    hiprz_node* nodes, uint32_t max_nodes, uint32_t* order)
        : m_boxes(boxes), m_leaf_size(leaf_size), m_root_leaf_size(root_leaf_size), m_nodes(nodes),
          m_max_nodes(max_nodes), m_order(order) {}

    bool build(const Box& root_box, std::vector<uint32_t>& items) {
        m_n_nodes = 1;
        m_n_order = 0;
        m_overflow = false;
        open(0, root_box, items.data(), items.data() + items.size(), 0);
        return !m_overflow;
    }
    uint32_t nodeCount() const { return m_n_nodes; }
    uint32_t orderCount() const { return m_n_order; }

private:
    static constexpr uint32_t kMaxDepth = 31;  // bvh_tree_node.hpp:14

    void leaf(uint32_t slot, const uint32_t* begin, const uint32_t* end) {
        hiprz_node& n = m_nodes[slot];
        Box bb{};
        const uint32_t count = uint32_t(end - begin);
        n.begin = m_n_order;
        n.meta = count | HIPRZ_NODE_LEAF;
        for (const uint32_t* it = begin; it != end; ++it) {
            if (it == begin) bb = m_boxes[*it];
            else grow(bb, m_boxes[*it]);
            m_order[m_n_order++] = *it;
        }
        std::memcpy(n.bb_min, bb.mn, 12);
        std::memcpy(n.bb_max, bb.mx, 12);
    }

    void inner(uint32_t slot, uint32_t ptype, const Box& bb_first, uint32_t* b0, uint32_t* e0, const Box& bb_second,
               uint32_t* b1, uint32_t* e1, uint32_t depth) {
        if (m_n_nodes + 2 > m_max_nodes) {
            m_overflow = true;
            leaf(slot, b0, b0);
            return;
        }
        const uint32_t c = m_n_nodes;
        m_n_nodes += 2;
        open(c, bb_first, b0, e0, depth + 1);
        open(c + 1, bb_second, b1, e1, depth + 1);
        hiprz_node& n = m_nodes[slot];
        n.begin = c;
        n.meta = ptype << HIPRZ_NODE_PTYPE_SHIFT;
        // fitBoundingBox: first child's box extended by the second's
        Box bb;
        std::memcpy(bb.mn, m_nodes[c].bb_min, 12);
        std::memcpy(bb.mx, m_nodes[c].bb_max, 12);
        Box sb;
        std::memcpy(sb.mn, m_nodes[c + 1].bb_min, 12);
        std::memcpy(sb.mx, m_nodes[c + 1].bb_max, 12);
        grow(bb, sb);
        std::memcpy(n.bb_min, bb.mn, 12);
        std::memcpy(n.bb_max, bb.mx, 12);
    }

    // `bb` is the box handed down by the parent (the split half of ITS hand-me-down box),
    // which is what the "too large" test and the child boxes are measured against.
    void open(uint32_t slot, const Box& bb, uint32_t* begin, uint32_t* end, uint32_t depth) {
        const ptrdiff_t count = end - begin;
        if (depth > kMaxDepth || count <= ptrdiff_t(m_leaf_size) || (depth == 0 && count <= ptrdiff_t(m_root_leaf_size)))
            return leaf(slot, begin, end);

        const float sx = bb.mx[0] - bb.mn[0], sy = bb.mx[1] - bb.mn[1], sz = bb.mx[2] - bb.mn[2];
        uint32_t* size_split = std::partition(begin, end, [&](uint32_t i) {
            const Box& o = m_boxes[i];
            return (o.mx[0] - o.mn[0]) < sx && (o.mx[1] - o.mn[1]) < sy && (o.mx[2] - o.mn[2]) < sz;
        });
        const ptrdiff_t n_split = size_split - begin, n_large = end - size_split;
        if (n_split != 0 && n_large != 0) return inner(slot, 3u, bb, begin, size_split, bb, size_split, end, depth);
        if (n_split == 0) return leaf(slot, size_split, end);

        float sp[3] = {0.0f, 0.0f, 0.0f};  // running mean of centroids
        for (ptrdiff_t i = 0; i < n_split; ++i)
            for (int a = 0; a < 3; ++a) sp[a] += (centroid(m_boxes[begin[i]], a) - sp[a]) / float(i + 1);
        float var[3] = {0.0f, 0.0f, 0.0f};
        uint32_t below[3] = {0, 0, 0};
        for (ptrdiff_t i = 0; i < n_split; ++i)
            for (int a = 0; a < 3; ++a) {
                const float c = centroid(m_boxes[begin[i]], a);
                const float d = c - sp[a];
                var[a] += d * d;
                below[a] += uint32_t(c < sp[a]);
            }
        if (!below[0] && !below[1] && !below[2]) return leaf(slot, begin, size_split);

        const float score[3] = {var[0] / float(n_split), var[1] / float(n_split), var[2] / float(n_split)};
        int axis = 2;
        if (score[0] >= score[1] && score[0] >= score[2] && below[0]) axis = 0;
        else if (score[1] >= score[0] && score[1] >= score[2] && below[1]) axis = 1;

        const float plane = sp[axis];
        uint32_t* mid = std::partition(begin, size_split, [&](uint32_t i) { return centroid(m_boxes[i], axis) < plane; });
        float hi[3] = {bb.mx[0], bb.mx[1], bb.mx[2]}, lo[3] = {bb.mn[0], bb.mn[1], bb.mn[2]};
        hi[axis] = lo[axis] = plane;
        static const uint32_t ptype_of_axis[3] = {2u, 1u, 0u};  // X=2, Y=1, Z=0 (bvh_tree_node.hpp:22-28)
        inner(slot, ptype_of_axis[axis], box_of(bb.mn, hi), begin, mid, box_of(lo, bb.mx), mid, size_split, depth);
    }

    const std::vector<Box>& m_boxes;
    uint32_t m_leaf_size, m_root_leaf_size;
    hiprz_node* m_nodes;
    uint32_t m_max_nodes;
    uint32_t* m_order;
    uint32_t m_n_nodes = 0, m_n_order = 0;
    bool m_overflow = false;
};

inline void normalize3(float* v) {
    const float s = 1.0f / std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    v[0] *= s, v[1] *= s, v[2] *= s;
}

inline bool ids_used(const uint32_t* ids) {
    return ids && !(ids[0] == 0xFFFFFFFFu && ids[1] == 0xFFFFFFFFu && ids[2] == 0xFFFFFFFFu);
}

// Math::vec3 rotations as the reference's CUDA restatement spells them
// (RayZath/cuda_render_parts.cuh:116-139).
struct V {
    float x, y, z;
};
inline V rx(V v, float a) {
    const float s = std::sin(a), c = std::cos(a);
    return {v.x, v.y * c + v.z * s, v.y * -s + v.z * c};
}
inline V ry(V v, float a) {
    const float s = std::sin(a), c = std::cos(a);
    return {v.x * c - v.z * s, v.y, v.x * s + v.z * c};
}
inline V rz(V v, float a) {
    const float s = std::sin(a), c = std::cos(a);
    return {v.x * c + v.y * s, v.x * -s + v.y * c, v.z};
}
inline void put(float* o, V v) { o[0] = v.x, o[1] = v.y, o[2] = v.z; }

inline uint32_t mix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}

}  // namespace

extern "C" {

int hiprz_build_mesh_tree(const hiprz_mesh_desc* mesh, hiprz_node* nodes_out, uint32_t max_nodes,
                          uint32_t* n_nodes_out, hiprz_tri* tris_out, hiprz_tri_attr* attrs_out) {
    if (!mesh || !nodes_out || !n_nodes_out || max_nodes < 1) return HIPRZ_ERR_INVALID;
    const uint32_t T = mesh->n_triangles;
    if (T && (!mesh->vertices || !mesh->tri_vertices || !tris_out || !attrs_out)) return HIPRZ_ERR_INVALID;
    for (uint32_t i = 0; i < 3 * T; ++i)
        if (mesh->tri_vertices[i] >= mesh->n_vertices) return HIPRZ_ERR_INVALID;

    std::vector<Box> boxes(T);
    std::vector<uint32_t> items(T);
    Box root{};
    for (uint32_t t = 0; t < T; ++t) {  // Triangle::boundingBox, mesh_component.cpp:27-33
        const float* p1 = mesh->vertices + 3 * mesh->tri_vertices[3 * t];
        const float* p2 = mesh->vertices + 3 * mesh->tri_vertices[3 * t + 1];
        const float* p3 = mesh->vertices + 3 * mesh->tri_vertices[3 * t + 2];
        boxes[t] = box_of(p1, p2);
        grow(boxes[t], p3);
        items[t] = t;
        if (t == 0) root = boxes[0];
        grow(root, boxes[t]);
    }
    std::vector<uint32_t> order(T ? T : 1);
    FlatTreeBuilder builder(boxes, 8u, 32u, nodes_out, max_nodes, order.data());
    if (!builder.build(root, items)) return HIPRZ_ERR_INVALID;
    *n_nodes_out = builder.nodeCount();
    return hiprz_fill_triangles(mesh, order.data(), T, tris_out, attrs_out);
}

int hiprz_fill_triangles(const hiprz_mesh_desc* mesh, const uint32_t* order, uint32_t n, hiprz_tri* tris_out, hiprz_tri_attr* attrs_out) {
    if (!mesh || (n && (!order || !tris_out || !attrs_out || !mesh->vertices || !mesh->tri_vertices))) return HIPRZ_ERR_INVALID;
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t t = order[i];
        if (t >= mesh->n_triangles) return HIPRZ_ERR_INVALID;
        for (int k = 0; k < 3; ++k)
            if (mesh->tri_vertices[3 * t + k] >= mesh->n_vertices) return HIPRZ_ERR_INVALID;
        hiprz_tri& o = tris_out[i];
        hiprz_tri_attr& a = attrs_out[i];
        std::memset(&o, 0, sizeof o);
        std::memset(&a, 0, sizeof a);
        const float* p1 = mesh->vertices + 3 * mesh->tri_vertices[3 * t];
        const float* p2 = mesh->vertices + 3 * mesh->tri_vertices[3 * t + 1];
        const float* p3 = mesh->vertices + 3 * mesh->tri_vertices[3 * t + 2];
        std::memcpy(o.v1, p1, 12);
        std::memcpy(o.v2, p2, 12);
        std::memcpy(o.v3, p3, 12);
        o.source_index = t;
        uint32_t flags = mesh->tri_materials ? (mesh->tri_materials[t] & HIPRZ_TRI_MATERIAL_MASK) : 0u;
        const uint32_t* tt = mesh->tri_texcrds ? mesh->tri_texcrds + 3 * t : nullptr;
        const uint32_t* tn = mesh->tri_normals ? mesh->tri_normals + 3 * t : nullptr;
        if (ids_used(tt)) {
            for (int k = 0; k < 3; ++k)
                if (tt[k] >= mesh->n_texcrds) return HIPRZ_ERR_INVALID;
            flags |= HIPRZ_TRI_HAS_TEXCRDS;
            std::memcpy(a.t1, mesh->texcrds + 2 * tt[0], 8);
            std::memcpy(a.t2, mesh->texcrds + 2 * tt[1], 8);
            std::memcpy(a.t3, mesh->texcrds + 2 * tt[2], 8);
        }
        if (ids_used(tn)) {
            for (int k = 0; k < 3; ++k)
                if (tn[k] >= mesh->n_normals) return HIPRZ_ERR_INVALID;
            flags |= HIPRZ_TRI_HAS_NORMALS;
            std::memcpy(a.n1, mesh->normals + 3 * tn[0], 12);
            std::memcpy(a.n2, mesh->normals + 3 * tn[1], 12);
            std::memcpy(a.n3, mesh->normals + 3 * tn[2], 12);
        }
        o.material_flags = flags;
        // Triangle::calculateNormal: normalize(cross(v2 - v3, v2 - v1)), mesh_component.cpp:19-26
        const float ax = p2[0] - p3[0], ay = p2[1] - p3[1], az = p2[2] - p3[2];
        const float bx = p2[0] - p1[0], by = p2[1] - p1[1], bz = p2[2] - p1[2];
        a.face_normal[0] = ay * bz - az * by;
        a.face_normal[1] = az * bx - ax * bz;
        a.face_normal[2] = ax * by - ay * bx;
        normalize3(a.face_normal);
    }
    return HIPRZ_OK;
}

int hiprz_build_world_tree(const hiprz_instance* instances, const uint8_t* has_mesh, uint32_t n_instances,
                           hiprz_node* nodes_out, uint32_t max_nodes, uint32_t* n_nodes_out,
                           uint32_t* order_out, uint32_t* n_order_out) {
    if (!nodes_out || !n_nodes_out || !n_order_out || max_nodes < 1) return HIPRZ_ERR_INVALID;
    if (n_instances && (!instances || !has_mesh || !order_out)) return HIPRZ_ERR_INVALID;
    // ObjectContainerWithBVH::update (bvh.hpp:29-53): the box starts from instance 0 (with
    // or without mesh) and grows by every instance that has one.
    std::vector<Box> boxes(n_instances);
    std::vector<uint32_t> items;
    Box root{};
    for (uint32_t i = 0; i < n_instances; ++i) {
        std::memcpy(boxes[i].mn, instances[i].bb_min, 12);
        std::memcpy(boxes[i].mx, instances[i].bb_max, 12);
    }
    if (n_instances) root = boxes[0];
    for (uint32_t i = 0; i < n_instances; ++i)
        if (has_mesh[i]) {
            grow(root, boxes[i]);
            items.push_back(i);
        }
    std::vector<uint32_t> scratch(1);
    FlatTreeBuilder builder(boxes, 4u, 8u, nodes_out, max_nodes, n_instances ? order_out : scratch.data());
    if (!builder.build(root, items)) return HIPRZ_ERR_INVALID;
    *n_nodes_out = builder.nodeCount();
    *n_order_out = builder.orderCount();
    return HIPRZ_OK;
}

// Instance::calculateBoundingBox, instance.cpp:117-155 (instance outside any group)
int hiprz_instance_bounds(const float* vertices, uint32_t n_vertices, hiprz_instance* inst) {
    if (!inst || (n_vertices && !vertices)) return HIPRZ_ERR_INVALID;
    std::memset(inst->bb_min, 0, 12);
    std::memset(inst->bb_max, 0, 12);
    if (n_vertices == 0) return HIPRZ_OK;
    Box b{};
    for (uint32_t i = 0; i < n_vertices; ++i) {
        const float sx = vertices[3 * i] * inst->scale[0], sy = vertices[3 * i + 1] * inst->scale[1],
                    sz = vertices[3 * i + 2] * inst->scale[2];
        float p[3];
        for (int a = 0; a < 3; ++a) p[a] = inst->x_axis[a] * sx + inst->y_axis[a] * sy + inst->z_axis[a] * sz;
        if (i == 0) b = box_of(p, p);
        else grow(b, p);
    }
    for (int a = 0; a < 3; ++a) {
        inst->bb_min[a] = b.mn[a] + inst->position[a];
        inst->bb_max[a] = b.mx[a] + inst->position[a];
    }
    return HIPRZ_OK;
}

// CoordSystem::applyRotation: RotatedXYZ (render_parts.cpp:51-56)
void hiprz_axes_from_rotation(const float r[3], float x_axis[3], float y_axis[3], float z_axis[3]) {
    put(x_axis, rz(ry(rx({1, 0, 0}, r[0]), r[1]), r[2]));
    put(y_axis, rz(ry(rx({0, 1, 0}, r[0]), r[1]), r[2]));
    put(z_axis, rz(ry(rx({0, 0, 1}, r[0]), r[1]), r[2]));
}
// CoordSystem::lookAt: RotatedZ().RotatedX().RotatedY() (render_parts.cpp:57-62)
void hiprz_axes_look_at(const float r[3], float x_axis[3], float y_axis[3], float z_axis[3]) {
    put(x_axis, ry(rx(rz({1, 0, 0}, r[2]), r[0]), r[1]));
    put(y_axis, ry(rx(rz({0, 1, 0}, r[2]), r[0]), r[1]));
    put(z_axis, ry(rx(rz({0, 0, 1}, r[2]), r[0]), r[1]));
}

float hiprz_seed_value(uint32_t seed, uint32_t pass, uint32_t i) {
    uint32_t h = mix32(seed ^ mix32(pass + 0x9E3779B9u));
    h = mix32(h ^ (i * 0x85EBCA6Bu + 1u));
    return float(h >> 8) * (20.0f / 16777216.0f) - 10.0f;
}

void hiprz_abi_sizes(uint32_t out[13]) {
    const uint32_t v[13] = {sizeof(hiprz_node),   sizeof(hiprz_tri),      sizeof(hiprz_tri_attr),   sizeof(hiprz_instance),
                            sizeof(hiprz_material), sizeof(hiprz_texture), sizeof(hiprz_spot_light), sizeof(hiprz_direct_light),
                            sizeof(hiprz_scene),  sizeof(hiprz_camera),   sizeof(hiprz_config),     sizeof(hiprz_counters),
                            sizeof(hiprz_mesh_desc)};
    std::memcpy(out, v, sizeof v);
}

const char* hiprz_version(void) { return "hiprz 0.1 (gfx950)"; }

}  // extern "C"

// =======================================================================================
// Opt-in mesh trees of better quality (hiprz_set_tree).  The reference's builder splits at the median centroid of
// the axis of largest variance (bvh_tree_node.hpp:117-215); a surface-area-heuristic tree visits fewer boxes for the same hits.  The
// tree only decides which boxes and triangles a ray meets, never what it hits, so frames stay the same — as long as ties between
// equally distant triangles are resolved as the reference's visiting order would: every triangle keeps its position in the
// reference's leaf order ("refpos") and the walks compare (t, refpos).
// =======================================================================================
namespace hiprz_trees {

struct SahBuilder {
    static constexpr int kBins = 16;
    // measured on config D (trace kernel, us): leaf 4 / cost 1.2: 1 018; 8 / 2: 959; 8 / 4: 932; 8 / 8: 931; 16 / 4: 945; 16 / 8: 1 009; the
    // reference trees: 1 022.  A node step runs at ~20 of 64 lanes, a triangle test in the cooperative phase at ~50, hence the high cost.
    uint32_t kMaxLeaf = 8;     // two quad entries of the cooperative triangle phase
    float kTraversal = 4.0f;   // cost of a node step relative to a triangle test
    const hiprz_tri* tris;                      // the mesh's triangles (global array), vertices v1 v2 v3
    std::vector<Box> bounds;
    std::vector<float> cx, cy, cz;
    std::vector<uint32_t> idx;                  // local triangle ids, permuted in place
    std::vector<hiprz_node>& nodes;
    uint32_t tri_base;                          // where this mesh's leaves start in the NEW triangle order

    SahBuilder(const hiprz_tri* t, uint32_t n, std::vector<hiprz_node>& out, uint32_t base) : tris(t), nodes(out), tri_base(base) {
        if (const char* e = std::getenv("HIPRZ_SAH_LEAF")) kMaxLeaf = uint32_t(std::atoi(e));
        if (const char* e = std::getenv("HIPRZ_SAH_TRAV")) kTraversal = float(std::atof(e));
        bounds.resize(n), cx.resize(n), cy.resize(n), cz.resize(n), idx.resize(n);
        for (uint32_t i = 0; i < n; ++i) {
            Box b = box_of(t[i].v1, t[i].v2);
            grow(b, t[i].v3);
            bounds[i] = b;
            cx[i] = centroid(b, 0), cy[i] = centroid(b, 1), cz[i] = centroid(b, 2);
            idx[i] = i;
        }
    }
    static float area(const Box& b) {
        const float dx = b.mx[0] - b.mn[0], dy = b.mx[1] - b.mn[1], dz = b.mx[2] - b.mn[2];
        return dx * dy + dy * dz + dz * dx;
    }
    const std::vector<float>& axis(int a) const { return a == 0 ? cx : a == 1 ? cy : cz; }

    // returns the index of the mesh's root in `nodes`
    uint32_t build() {
        struct Item {
            uint32_t node, first, count, depth;
        };
        const uint32_t root = uint32_t(nodes.size());
        nodes.push_back(hiprz_node{});
        std::vector<Item> stack{{root, 0u, uint32_t(idx.size()), 1u}};
        while (!stack.empty()) {
            const Item it = stack.back();
            stack.pop_back();
            Box box = bounds[idx[it.first]], cbox;
            for (int a = 0; a < 3; ++a) cbox.mn[a] = cbox.mx[a] = axis(a)[idx[it.first]];
            for (uint32_t k = 1; k < it.count; ++k) {
                const uint32_t t = idx[it.first + k];
                grow(box, bounds[t]);
                const float c[3] = {cx[t], cy[t], cz[t]};
                grow(cbox, c);
            }
            hiprz_node& node = nodes[it.node];
            std::memcpy(node.bb_min, box.mn, 12), std::memcpy(node.bb_max, box.mx, 12);
            auto make_leaf = [&]() {
                std::sort(idx.begin() + it.first, idx.begin() + it.first + it.count);  // ascending refpos inside a leaf
                nodes[it.node].begin = tri_base + it.first;
                nodes[it.node].meta = HIPRZ_NODE_LEAF | it.count;
            };
            if (it.count <= 2u || it.depth >= 60u) {
                make_leaf();
                continue;
            }
            // binned SAH over the three axes
            float best_cost = 3.4e38f;
            int best_axis = -1, best_plane = 0;
            for (int a = 0; a < 3; ++a) {
                const float lo = cbox.mn[a], extent = cbox.mx[a] - cbox.mn[a];
                if (!(extent > 0.0f)) continue;
                const float scale = float(kBins) / extent;
                Box bin_box[kBins];
                uint32_t bin_count[kBins] = {0};
                const std::vector<float>& c = axis(a);
                for (uint32_t k = 0; k < it.count; ++k) {
                    const uint32_t t = idx[it.first + k];
                    int b = int((c[t] - lo) * scale);
                    b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
                    if (bin_count[b]++ == 0) bin_box[b] = bounds[t];
                    else grow(bin_box[b], bounds[t]);
                }
                float right_area[kBins];
                uint32_t right_count[kBins];
                Box acc{};
                uint32_t n = 0;
                for (int b = kBins - 1; b > 0; --b) {
                    if (bin_count[b]) {
                        if (n == 0) acc = bin_box[b];
                        else grow(acc, bin_box[b]);
                        n += bin_count[b];
                    }
                    right_area[b] = n ? area(acc) : 0.0f, right_count[b] = n;
                }
                n = 0;
                for (int b = 0; b < kBins - 1; ++b) {
                    if (bin_count[b]) {
                        if (n == 0) acc = bin_box[b];
                        else grow(acc, bin_box[b]);
                        n += bin_count[b];
                    }
                    if (n == 0 || right_count[b + 1] == 0) continue;
                    const float cost = area(acc) * float(n) + right_area[b + 1] * float(right_count[b + 1]);
                    if (cost < best_cost) best_cost = cost, best_axis = a, best_plane = b + 1;
                }
            }
            const float node_area = area(box);
            const float split_cost = best_axis >= 0 && node_area > 0.0f ? kTraversal + best_cost / node_area : 3.4e38f;
            if (it.count <= kMaxLeaf && float(it.count) <= split_cost) {
                make_leaf();
                continue;
            }
            uint32_t mid;
            int split_axis = best_axis;
            if (best_axis >= 0 && (it.count > kMaxLeaf || split_cost < float(it.count))) {
                const float lo = cbox.mn[best_axis], scale = float(kBins) / (cbox.mx[best_axis] - cbox.mn[best_axis]);
                const std::vector<float>& c = axis(best_axis);
                auto first = idx.begin() + it.first;
                mid = uint32_t(std::partition(first, first + it.count, [&](uint32_t t) {
                          int b = int((c[t] - lo) * scale);
                          b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
                          return b < best_plane;
                      }) - first);
            } else {  // every centroid in one spot (or no plane separates): halve along the widest axis of the box
                split_axis = 0;
                for (int a = 1; a < 3; ++a)
                    if (box.mx[a] - box.mn[a] > box.mx[split_axis] - box.mn[split_axis]) split_axis = a;
                const std::vector<float>& c = axis(split_axis);
                auto first = idx.begin() + it.first;
                std::nth_element(first, first + it.count / 2, first + it.count, [&](uint32_t x, uint32_t y) { return c[x] < c[y] || (c[x] == c[y] && x < y); });
                mid = it.count / 2;
            }
            if (mid == 0u || mid == it.count) {
                make_leaf();
                continue;
            }
            const uint32_t children = uint32_t(nodes.size());
            nodes.push_back(hiprz_node{}), nodes.push_back(hiprz_node{});
            // partition type of the node layout: X = 2, Y = 1, Z = 0 (bvh_tree_node.hpp:22-28); the first child holds the lower centroids
            nodes[it.node].begin = children;
            nodes[it.node].meta = uint32_t(2 - split_axis) << HIPRZ_NODE_PTYPE_SHIFT;
            stack.push_back({children + 1u, it.first + mid, it.count - mid, it.depth + 1u});
            stack.push_back({children, it.first, mid, it.depth + 1u});
        }
        return root;
    }
};

}  // namespace hiprz_trees

extern "C" int hiprz_rebuild_mesh_trees(const hiprz_scene* sc, uint32_t method, hiprz_node* nodes_out, uint32_t max_nodes, uint32_t* n_nodes_out,
                                        uint32_t* tri_order_out, uint32_t* blas_roots_out, uint32_t* tlas_root_out) {
    if (!sc || !nodes_out || !n_nodes_out || !tri_order_out || !blas_roots_out || !tlas_root_out || (method != 1u && method != 2u)) return HIPRZ_ERR_INVALID;
    std::vector<hiprz_node> nodes;
    // the world tree is copied as it is (breadth-first, children adjacent)
    *tlas_root_out = 0u;
    if (sc->n_instances && sc->n_tlas_order) {
        if (sc->tlas_root >= sc->n_nodes) return HIPRZ_ERR_INVALID;
        std::vector<uint32_t> queue{sc->tlas_root};
        nodes.push_back(sc->nodes[sc->tlas_root]);
        for (size_t q = 0; q < queue.size(); ++q) {
            const hiprz_node old = sc->nodes[queue[q]];
            if (old.meta & HIPRZ_NODE_LEAF) continue;
            if (uint64_t(old.begin) + 1 >= sc->n_nodes || nodes.size() > size_t(2) * sc->n_nodes) return HIPRZ_ERR_INVALID;
            nodes[q].begin = uint32_t(nodes.size());
            queue.push_back(old.begin), queue.push_back(old.begin + 1u);
            nodes.push_back(sc->nodes[old.begin]), nodes.push_back(sc->nodes[old.begin + 1u]);
        }
    }
    // every distinct mesh: its triangle range from the leaves of its reference tree, then a new tree over that range
    std::vector<uint32_t> new_root(sc->n_nodes, 0xFFFFFFFFu);
    uint32_t tri_cursor = 0u;
    std::vector<uint8_t> tri_seen(sc->n_tris, 0);
    // an instance without a mesh is in no leaf of the world tree (bvh.hpp:40-47) and is never entered: whatever its blas_root field holds
    // (the hosts leave it 0, which is a node of the WORLD tree) is not a mesh
    std::vector<uint8_t> in_world(sc->n_instances, 0);
    for (uint32_t k = 0; k < sc->n_tlas_order; ++k)
        if (sc->tlas_order[k] < sc->n_instances) in_world[sc->tlas_order[k]] = 1;
    for (uint32_t i = 0; i < sc->n_instances; ++i) {
        const uint32_t root = sc->instances[i].blas_root;
        blas_roots_out[i] = in_world[i] ? root : 0xFFFFFFFFu;
        if (!in_world[i] || root >= sc->n_nodes) continue;
        if (new_root[root] == 0xFFFFFFFFu) {
            uint32_t lo = 0xFFFFFFFFu, total = 0u;
            std::vector<uint32_t> walk{root};
            size_t steps = 0;
            while (!walk.empty()) {
                if (++steps > size_t(sc->n_nodes) + 1u) return HIPRZ_ERR_INVALID;
                const hiprz_node n = sc->nodes[walk.back()];
                walk.pop_back();
                if (n.meta & HIPRZ_NODE_LEAF) {
                    const uint32_t count = n.meta & HIPRZ_NODE_COUNT_MASK;
                    if (uint64_t(n.begin) + count > sc->n_tris) return HIPRZ_ERR_INVALID;
                    if (count) lo = std::min(lo, n.begin), total += count;
                } else {
                    if (uint64_t(n.begin) + 1 >= sc->n_nodes) return HIPRZ_ERR_INVALID;
                    walk.push_back(n.begin), walk.push_back(n.begin + 1u);
                }
            }
            if (total == 0u) {
                new_root[root] = uint32_t(nodes.size());
                hiprz_node leaf = sc->nodes[root];
                leaf.begin = 0u, leaf.meta = HIPRZ_NODE_LEAF;
                nodes.push_back(leaf);
            } else {
                if (uint64_t(lo) + total > sc->n_tris) return HIPRZ_ERR_INVALID;
                for (uint32_t t = lo; t < lo + total; ++t) {
                    if (tri_seen[t]) return HIPRZ_ERR_INVALID;  // the leaves of a mesh must tile one contiguous range
                    tri_seen[t] = 1;
                }
                if (method == 2u) {  // HIPRZ_TREE_DEVICE: one leaf over the whole mesh — what the device build starts from
                    new_root[root] = uint32_t(nodes.size());
                    hiprz_node leaf = sc->nodes[root];
                    leaf.begin = tri_cursor, leaf.meta = HIPRZ_NODE_LEAF | total;
                    nodes.push_back(leaf);
                    for (uint32_t k = 0; k < total; ++k) tri_order_out[tri_cursor + k] = lo + k;
                } else {
                    hiprz_trees::SahBuilder builder(sc->tris + lo, total, nodes, tri_cursor);
                    new_root[root] = builder.build();
                    for (uint32_t k = 0; k < total; ++k) tri_order_out[tri_cursor + k] = lo + builder.idx[k];
                }
                tri_cursor += total;
            }
        }
        blas_roots_out[i] = new_root[root];
    }
    for (uint32_t t = 0; t < sc->n_tris; ++t)  // triangles no instance reaches keep a slot behind the others
        if (!tri_seen[t]) tri_order_out[tri_cursor++] = t;
    if (nodes.size() > max_nodes) return HIPRZ_ERR_INVALID;
    std::memcpy(nodes_out, nodes.data(), nodes.size() * sizeof(hiprz_node));
    *n_nodes_out = uint32_t(nodes.size());
    return HIPRZ_OK;
}
