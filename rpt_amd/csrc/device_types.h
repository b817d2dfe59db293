// device_types.h — the flattened, device-resident scene and the wavefront path state.
// Shared by the host flattener / API (host_scene.cpp, api_*.cpp) and the gfx950 kernels.
//
// Layout rules (MI355X): everything a lane gathers on its own (kd nodes, triangles, group
// children) is a 16-byte-aligned record so one `global_load_dwordx4` (or a run of them) fetches
// it; everything a whole wave reads at the same index (top-level objects, lights, materials,
// trees) is read through uniform addresses and lands in SGPRs via the scalar cache; per-path
// state is structure-of-arrays indexed by path slot so that consecutive lanes touch
// consecutive 8-byte words.
#pragma once
#include <stdint.h>

namespace rptdev {

constexpr int KD_MAX_STACK = 32; // deepest kd-tree the fast traversals' stacks hold; deeper ones are walked by rpt_tree_generic
// Light::Object whose shape is a group: Shape::sample descends through this many group levels at most (kernels/
// sampling.inc keeps the chain for the way back in registers).  Geometry nests without limit (rpt_tree_generic).
#ifndef RPT_MAX_NEST
#define RPT_MAX_NEST 7
#endif
constexpr int KD_LDS_LEVELS = 12;   // stack levels the persistent kernel keeps in LDS (rest: scratch)
#ifndef RPT_KD_LDS_LEVELS_WF
#define RPT_KD_LDS_LEVELS_WF 10
#endif
constexpr int KD_LDS_LEVELS_WF = RPT_KD_LDS_LEVELS_WF; // same for the wavefront kernels (3 blocks of 256 threads per CU share 160 KB)

// One kd node: 16 B, one dwordx4 load.  Inner: a = left child (right = a+1), ib = axis (0..2).
// Leaf: a = first entry in refs[], ib = 3 | (count << 2).
struct alignas(16) KdNode {
  double split;
  uint32_t a;
  uint32_t ib;
};

// Triangle exactly as the reference stores it (mesh.rs:8-22): v1 v2 v3 n1 n2 n3, 144 B.
struct alignas(16) Tri {
  double v[18];
};

// The per-triangle quantities Triangle::intersect re-derives on every test (mesh.rs:50-51,64-69),
// computed once on the host with the same IEEE operations (so the bits are the ones the
// reference computes): plane normal, edge vectors and their dot products.  128 B = 8 dwordx4.
struct alignas(16) TriX {
  double pn[3];  // normalize((v2-v1) x (v3-v1))
  double v1[3];
  double d0[3];  // v2 - v1
  double d1[3];  // v3 - v1
  double d00, d01, d11, denom;
};

// Conservative bounding box of one leaf entry of a MESH tree, in leaf order like TriX: 16-bit fixed-point
// coordinates relative to the tree's bounds (Tree::qlo / qscale), minima rounded down and maxima rounded up by
// one step more than needed, stored per axis as CENTRE and HALF-EXTENT — the slab interval of an axis is then
// t(centre) -+ half * |a|, whatever the sign of the direction: one fma and one packed fma instead of two fma's,
// a min and a max.  16 B = ONE per-lane gather.  It is a FILTER in front of Triangle::intersect: a
// ray that misses the box (in the window [t_min, record.time]) cannot be accepted by the exact test, so skipping
// the exact test changes nothing (kernels/shapes.inc, leaf_box_pass).
// entries of padding at the end of the LeafBox array: the filter's batches over a GROUP leaf (4 boxes) read unclamped — one
// address and immediates; the bits past the leaf's end are masked out
#define RPT_LBOX_PAD 16
struct alignas(16) LeafBox {
  uint32_t w[4]; // w[k] = centre_k | half_k << 16 for k = x, y, z; w3: 1 = the whole grid (never filtered)
};

// A placed shape: a top-level scene object, a light's shape, or a child of a GROUP tree.
struct alignas(16) Inst {
  int32_t kind;     // RPT_SHAPE_*
  int32_t has_xf;   // Transformed<T> wrapper present
  int32_t tree;     // MESH / GROUP: index into trees[]
  int32_t material; // top-level objects: index into materials[]
  double inv[16];   // inverse_transform, column-major (shape.rs:105)
  double nrm[9];    // normal_transform, column-major (shape.rs:106)
  double fwd[16];   // transform (shape.rs:103)      — sampling only
  double lin[9];    // linear (shape.rs:104)         — sampling only
  double scale;     // determinant (shape.rs:107)    — sampling only
  double plane[4];  // PLANE: normal xyz, value
  double bounds[6]; // MESH / GROUP: a copy of trees[tree].bounds, so that a root slab test needs one scalar
                    // round trip (this record) instead of two (record, then tree)
  // flat scenes: where the six slab quotients of this (untransformed) mesh sit in the per-ray plane table —
  // 4 bits per face in the order bounds[0], [3], [1], [4], [2], [5]; slot = axis * 4 + index (kernels/paths.inc)
  uint32_t plane_idx;
  uint32_t plane_use;
};

struct alignas(16) Tree {
  uint32_t node_base; // into nodes[]
  uint32_t ref_base;  // into refs[]
  uint32_t prim_base; // into tris[] (MESH) or insts[] (GROUP)
  uint32_t num_prims;
  double bounds[6];   // p_min xyz, p_max xyz (kdtree.rs:103)
  uint32_t regular;   // 1: every split lies inside its cell -> the compact traversal is exact
  uint32_t root_leaf; // 0, or 1 + the entry count when node 0 is a leaf (fewer than 16 primitives, kdtree.rs:236)
  uint32_t root_first; // that leaf's first entry in refs[] / lrec[]
  uint32_t split_range_ok; // regular, and every split value is 0 or has 2^-340 <= |v| < 2^399 (rpt_tree_trace's fast division)
  uint64_t sample_zone; // rand 0.8 UniformInt zone for Uniform::from(0..num_prims): u64::MAX - (2^64 - n) % n,
                        // precomputed because a 64-bit modulo costs ~200 device instructions per light sample
  uint32_t mesh_kids;   // GROUP: some child is a MESH (a kd-tree of kd-trees): such an object is walked by the per-tree
                        // kernels whatever its own depth (api_scene.cpp)
  uint32_t generic_only; // the tree of a top-level object that only rpt_tree_generic walks (a group among a group's
                        // children, mesh children rpt_nest_trace does not take): rpt_tree_enter hands it every ray.
                        // (A tree deeper than KD_MAX_STACK is NOT one: rpt_tree_trace takes it, with the levels
                        // beyond its LDS stack in the spill columns, which api_render.cpp sizes from the deepest tree.)
  double qlo[3];        // MESH: origin and step of the LeafBox fixed-point grid (coordinate = qlo + q * qscale)
  double qscale[3];
};

struct alignas(16) Material {
  double color[3];
  double index, roughness, metallic, emittance;
  int32_t transparent;
  int32_t _pad;
};

struct alignas(16) Light {
  int32_t kind; // RPT_LIGHT_*
  int32_t inst; // OBJECT: index into insts[] of the light's shape
  double color[3];
  double vec[3];
  double mat_color[3]; // OBJECT: material.color
  double mat_emittance;
};

struct Scene {
  const Inst* insts;
  const Tree* trees;
  const KdNode* nodes;
  const uint32_t* refs;
  const Tri* tris;   // vertices + vertex normals (sampling, shading normals)
  const TriX* lrec;  // intersection-ready records in LEAF order: lrec[j] belongs to refs[j]
  const LeafBox* lbox; // conservative boxes of the same entries (MESH trees), leaf order
  const Material* materials;
  const Light* lights;
  const double* env_texels; // HDRI: width*height*3
  double env_color[3];
  uint32_t env_width, env_height;
  int32_t env_kind;
  int32_t num_objects; // insts[0..num_objects) are scene.objects in order
  int32_t num_lights;
  int32_t num_shadow_lights; // lights that cast a shadow ray (non-ambient)
  int32_t force_general;     // tests: always take the general (box-carrying) traversal
  int32_t use_leaf_boxes;    // conservative box filter in front of Triangle::intersect (RPTGPU_LEAF_BOXES, default on)
};

// camera constants, precomputed on the host exactly as Camera::cast_ray derives them per call
// (camera.rs:66-67): d = 1/tan(fov/2), right = normalize(direction x up)
struct Camera {
  double eye[3], direction[3], up[3], right[3];
  double d;
  double aperture, focal_distance;
};

// Per-pass wavefront state, SoA over path slots (slot = s_local * npix + pixel_local).
// REC_FIELDS doubles per depth record: A[3], f[3], 1/pdf, |wi.n|  (the nested clamp of
// renderer.rs:162-167 is folded back-to-front by the resolve kernel, along the records' parent links).
constexpr int REC_FIELDS = 8;
constexpr int SHADOW_FIELDS = 7; // wi[3], dist, contribution[3]

// Round 6: the state of a depth is DENSE — index i is the path's position among the paths alive at this depth, not a
// slot it keeps for life.  rpt_shade reads ray / hit / draw / pid / col at i and writes the survivors' state to the *_next
// arrays at their position j in the next depth's queue (the host swaps the pairs), so every depth's kernels stream
// contiguous state; until then the arrays were indexed by the path's slot through a queue of slots, and from the second
// depth on every access was a scattered 8-byte gather (rpt_shade: 2.2 x the time per path of depth 0).
struct PathState {
  double* ray;       // [6][cap]   ox oy oz dx dy dz
  double* hit;       // [4][cap]   t nx ny nz
  int32_t* hit_obj;  // [cap]      object index or -1
  uint32_t* draw;    // [cap]      Philox draw counter of the path's stream
  uint32_t* pid;     // [cap]      which path: sample_local * npix + pixel_local (its Philox stream, its place in the frame)
  uint32_t* col;     // [cap]      the column of the path's record one depth up (depth >= 1)
  double* ray_next;  // the same four for the next depth, written by rpt_shade at the survivors' positions
  uint32_t* draw_next;
  uint32_t* pid_next;
  uint32_t* col_next;
  double* rec;       // [rec_cap][REC_FIELDS]: the pass's depth records, one 64-byte COLUMN per (path, depth) the path reached — the
                     // columns of depth d are [rec_off_d, rec_off_d + n_active_d) in the order of that depth's queue (round
                     // 6; until then [max_bounces + 1][REC_FIELDS][cap]: 1 088 B per path at 16 bounces whatever its length)
  uint32_t* rec_parent; // [rec_cap] the column of the same path's record one depth up (REC_NONE at depth 0)
  uint32_t* last_col;   // [cap] by path id: the column of the record the path ENDED with (written once, by its last rpt_shade)
  double* shadow;    // [num_lights][SHADOW_FIELDS][cap]
  uint64_t cap;      // slots allocated (stride of every per-path array above)
  uint64_t rec_cap;  // columns allocated
  uint32_t* sort_keys; // path re-order (in-kernel-traversal scenes; nullptr = off): rpt_shade writes the survivor's ray key
  uint32_t* sort_vals; // and its position here
  double* next_rows;   // ... and its next state as ONE 64-byte row [cap][8] (o, d, draw | pid, col | 0) instead of the *_next
                       // arrays: rpt_path_permute gathers rows in sorted order — one sector per path, not nine
  double key_bounds[6]; // the key's grid: the bounds of the scene's bounded objects
};
constexpr uint32_t REC_NONE = 0xffffffffu;

struct Frame {
  uint32_t width, height;
  uint32_t npix;            // pixels assigned to this part
  const uint32_t* pixels;   // [npix] pixel indices y*width+x, ascending
  uint32_t max_bounces;
  uint32_t _pad;
  uint64_t seed;
  uint64_t sample_base;     // global index of the first sample of this pass
  double* accum;            // [npix][3] running sum of L_0 over samples
};

} // namespace rptdev
