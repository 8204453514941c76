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

    for (uint32_t i = 0; i < T; ++i) {
        const uint32_t t = order[i];
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
