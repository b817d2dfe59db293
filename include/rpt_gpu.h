/*
 * rpt_gpu.h — C ABI of the MI355X (gfx950) path-tracing back-end for the `rpt` renderer.
 *
 * This is the drop-in boundary for rpt's one hot path: the body of
 * `Renderer::sample(&self, iterations, &mut Buffer)` (reference src/renderer.rs:117-129),
 * which both public entry points call (`render` renderer.rs:96-100, `iterative_render`
 * renderer.rs:103-115).  The reference has no FFI of its own (`#![forbid(unsafe_code)]`,
 * src/lib.rs:3); the entry points below are what an `rpt-gpu-sys` style binding would bind
 * (see INTEGRATION.md for the Rust / ctypes / C++ stubs).
 *
 * Conventions
 *   - plain C, plain pointers and sizes, no C++/torch types;
 *   - every POD struct mirrors a reference struct field by field (cited per struct);
 *   - matrices are column-major, exactly as nalgebra/glm stores them
 *     (element (row r, col c) of a 4x4 is m[c*4+r], of a 3x3 is m[c*3+r]);
 *   - every function returns 0 (RPTGPU_OK) or a negative RPTGPU_E_* code, never throws,
 *     never aborts; rptgpu_strerror() / rptgpu_last_error_detail() explain;
 *   - the caller owns every input pointer and every output buffer for the duration of the
 *     call only; the library copies what it needs to device memory in rptgpu_scene_create;
 *   - a handle is NOT re-entrant (one render in flight per handle); distinct handles may be
 *     used from distinct threads.  All calls are synchronous unless they take a stream.
 */
#ifndef RPT_GPU_H
#define RPT_GPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RPTGPU_ABI_VERSION 7

/* ---- error codes (replace the reference's panics: buffer.rs:26,33,89, plane.rs:35) ---- */
enum {
  RPTGPU_OK = 0,
  RPTGPU_E_INVALID_ARGUMENT = -1,
  RPTGPU_E_UNSUPPORTED_SHAPE = -2, /* shape outside the closed device set          */
  RPTGPU_E_NO_DEVICE = -3,         /* no HIP device / HIP runtime failure at init              */
  RPTGPU_E_HIP = -4,               /* a HIP call failed; see rptgpu_last_error_detail          */
  RPTGPU_E_OUT_OF_MEMORY = -5,
  RPTGPU_E_TREE_TOO_DEEP = -6,     /* (not returned for any input since ABI v5; internal check)  */
  RPTGPU_E_UNIMPLEMENTED_SAMPLE = -7, /* Light::Object over a Plane: plane.rs:34-36 panics     */
  RPTGPU_E_COMM = -8               /* RCCL is not available or a collective failed             */
};

/* ---- Material: src/material.rs:8-26 ---- */
typedef struct RptMaterial {
  double color[3];
  double index;
  double roughness;
  double metallic;
  double emittance;
  int32_t transparent; /* bool */
  int32_t _pad;
} RptMaterial;

/* ---- Triangle: src/shape/mesh.rs:8-22 (three vertices, three normals) ---- */
typedef struct RptTriangle {
  double v1[3], v2[3], v3[3];
  double n1[3], n2[3], n3[3];
} RptTriangle;

/* ---- Transformed<T>: src/shape.rs:101-108, the five precomputed fields ---- */
typedef struct RptTransform {
  double transform[16];         /* M                                 shape.rs:103 */
  double linear[9];             /* mat4_to_mat3(M)                   shape.rs:104 */
  double inverse_transform[16]; /* glm::inverse(M)                   shape.rs:105 */
  double normal_transform[9];   /* glm::inverse_transpose(linear)    shape.rs:106 */
  double scale;                 /* linear.determinant()              shape.rs:107 */
} RptTransform;

/* The closed set of shapes the device understands (the reference's `dyn Shape` is open). */
enum {
  RPT_SHAPE_SPHERE = 0, /* src/shape/sphere.rs:9    unit sphere at the origin                 */
  RPT_SHAPE_PLANE = 1,  /* src/shape/plane.rs:7-13  x . normal = value                        */
  RPT_SHAPE_CUBE = 2,   /* src/shape/cube.rs:8      unit cube [-0.5,0.5]^3                    */
  RPT_SHAPE_MESH = 3,   /* src/shape/mesh.rs:102    Mesh = KdTree<Triangle>                   */
  RPT_SHAPE_GROUP = 4,  /* KdTree<Box<dyn Bounded>> (examples/fractal_spheres.rs:45,
                           fractal_teapots.rs:69).  Children: anything Bounded — SPHERE, CUBE, MESH,
                           MONOMIAL, or another GROUP — each optionally Transformed (a PLANE is not
                           Bounded, kdtree.rs:9-12, and is refused).  MESH children that share one
                           triangle array (Arc<Mesh>) share one tree on the device.  NESTING: GROUPs may
                           contain GROUPs to any depth, as in the reference (kdtree.rs:14-24 forwards
                           Bounded through Box).  A group with tree children is walked by the per-tree
                           kernels of the wavefront pipeline (two regular levels in one loop, anything else
                           by the generic walker); RPT_FLAG_PERSISTENT is ignored for such a scene.  Only a
                           Light::Object's shape keeps a limit: eight group levels (Shape::sample) */
  RPT_SHAPE_MONOMIAL = 5 /* src/shape/monomial_surface.rs:12-18  y = height*(x^2+z^2)^(exp/2),
                            x^2+z^2 <= 1; like the reference, intersection and normals are
                            only valid for exp = 4 (monomial_surface.rs:10): any other exp is
                            refused.  As a top-level object, a Light::Object, or a GROUP child  */
};

typedef struct RptShape {
  int32_t kind;        /* RPT_SHAPE_*                                                          */
  int32_t transformed; /* 1 = wrapped in Transformed<T> (shape.rs:101), xf is valid            */
  RptTransform xf;
  double plane_normal[3]; /* PLANE: plane.rs:9  */
  double plane_value;     /* PLANE: plane.rs:12 */
  double monomial_height; /* MONOMIAL: monomial_surface.rs:15 */
  double monomial_exp;    /* MONOMIAL: monomial_surface.rs:17 (must be 4) */
  const RptTriangle* triangles; /* MESH: KdTree<Triangle>::objects (kdtree.rs:102)             */
  uint64_t num_triangles;
  const struct RptShape* children; /* GROUP: KdTree<Box<dyn Bounded>>::objects                 */
  uint64_t num_children;
} RptShape;

/* ---- Object: src/object.rs:10-16 (one shape, one material) ---- */
typedef struct RptObject {
  RptShape shape;
  RptMaterial material;
} RptObject;

/* ---- Light: src/light.rs:7-19 ---- */
enum {
  RPT_LIGHT_POINT = 0,       /* Point(color, location)        */
  RPT_LIGHT_AMBIENT = 1,     /* Ambient(color)                */
  RPT_LIGHT_DIRECTIONAL = 2, /* Directional(color, direction) */
  RPT_LIGHT_OBJECT = 3       /* Object(Object)                */
};

typedef struct RptLight {
  int32_t kind;
  int32_t _pad;
  double color[3];
  double vec[3];    /* POINT: location; DIRECTIONAL: direction (not normalised by the caller) */
  RptObject object; /* OBJECT */
} RptLight;

/* ---- Environment: src/environment.rs:56-62, Hdri: environment.rs:5-15 ---- */
enum { RPT_ENV_COLOR = 0, RPT_ENV_HDRI = 1 };

typedef struct RptEnvironment {
  int32_t kind;
  int32_t _pad;
  double color[3];      /* COLOR */
  uint32_t width;       /* HDRI  */
  uint32_t height;      /* HDRI  */
  const double* texels; /* HDRI: width*height*3 doubles, row-major, row 0 = polar angle 0
                           (environment.rs:14 `buf: Vec<Color>`)                              */
} RptEnvironment;

/* ---- Scene: src/scene.rs:7-16 ---- */
typedef struct RptScene {
  const RptObject* objects;
  uint64_t num_objects;
  const RptLight* lights;
  uint64_t num_lights;
  RptEnvironment environment;
} RptScene;

/* ---- Camera: src/camera.rs:8-26 ---- */
typedef struct RptCamera {
  double eye[3];
  double direction[3];
  double up[3];
  double fov;
  double aperture;
  double focal_distance;
} RptCamera;

/* Arithmetic modes.  STRICT is the parity mode and the only one: IEEE f64, no FMA contraction, true
 * divisions — the arithmetic of the reference (Rust never contracts).  ABI versions <= 3 also had
 * F64_FAST = 1 (the same kernels with contraction allowed): it measured SLOWER than STRICT and was not
 * bit-exact, so it was removed; the value 1 is now refused with RPTGPU_E_INVALID_ARGUMENT.  */
enum {
  RPT_PRECISION_F64_STRICT = 0
};

enum {
  RPT_FLAG_PROFILE_KERNELS = 1u, /* bracket every kernel with HIP events (rptgpu_get_stats) */
  RPT_FLAG_GENERAL_TRAVERSAL = 4u, /* tests: always use the general (box-carrying, scratch-stack)
                                      kd traversal instead of the compact LDS-stack one */
  RPT_FLAG_WAVEFRONT = 2u,       /* force the multi-kernel wavefront pipeline (raygen / extend / shade /
                                    shadow / resolve; path state in HBM, lean 4-waves/SIMD traversal
                                    kernels with LDS stacks) */
  RPT_FLAG_PERSISTENT = 8u       /* force the persistent path kernel (whole path in registers).
                                    With neither flag the library picks: wavefront when the scene has
                                    real kd-trees (depth >= 3: traversal latency dominates and wants
                                    occupancy), persistent otherwise.  Both give the same bits. */
};

/* ---- what Renderer carries into sample(): src/renderer.rs:18-42 + the call argument ----
 * seed / sample_index_base are ADDITIONS: the reference seeds every row from OS entropy
 * (renderer.rs:121) and is not reproducible.  Random numbers are Philox4x32-10 keyed by
 * `seed`, counter (pixel index, sample index, draw block): the image does not depend on how
 * pixels are partitioned over devices. */
typedef struct RptRenderParams {
  uint32_t width;       /* renderer.rs:26 */
  uint32_t height;      /* renderer.rs:29 */
  uint32_t max_bounces; /* renderer.rs:38 */
  uint32_t iterations;  /* argument of sample(): paths per pixel in this batch (renderer.rs:117) */
  double exposure_value; /* renderer.rs:32 */
  uint64_t seed;
  uint64_t sample_index_base; /* global index of this batch's first sample                    */
  /* pixel partition for multi-GPU: the image is cut into tile_width x tile_height tiles,
     numbered row-major; this call renders tiles with (tile_id % part_count) == part_index and
     writes 0.0 to every other pixel, so the sum over parts is the full frame.
     part_count = 0 or 1 renders everything. */
  uint32_t tile_width;
  uint32_t tile_height;
  uint32_t part_index;
  uint32_t part_count;
  uint32_t precision_mode; /* RPT_PRECISION_* */
  uint32_t flags;          /* RPT_FLAG_*      */
  /* ABI v6 — how rptgpu_render_batch_reduce brings the ranks' pixels to the root (ignored by every other call):
     RPT_COLLECTIVE_GATHER (also 0): each rank sends only the pixels it owns (ncclSend / ncclRecv), RPT_COLLECTIVE_REDUCE:
     ncclReduce(sum) of full frames that are zero outside the rank's tiles.  Same frame either way.  Every rank of a
     batch must pass the same value.  Until v5: the environment variable RPTGPU_COLLECTIVE, which still overrides. */
  uint32_t collective;
  uint32_t _reserved0;     /* 0 */
} RptRenderParams;
enum { RPT_COLLECTIVE_DEFAULT = 0, RPT_COLLECTIVE_GATHER = 1, RPT_COLLECTIVE_REDUCE = 2 };

/* Per-kernel accounting since the last rptgpu_reset_stats (filled when
 * RPT_FLAG_PROFILE_KERNELS is set; counts are always maintained). */
enum {
  RPT_K_RAYGEN = 0,
  RPT_K_EXTEND = 1, /* closest-hit traversal + intersection      */
  RPT_K_SHADE = 2,  /* emission, NEE generation, BSDF sampling   */
  RPT_K_SHADOW = 3, /* shadow-ray traversal                      */
  RPT_K_RESOLVE = 4,/* nested firefly-clamp fold + accumulation  */
  RPT_K_PATHS = 5,  /* persistent path kernel: all of the above in registers */
  RPT_K_TREE_TRACE = 6, /* per-tree persistent kd traversal (closest-hit and shadow queries of deep trees);
                           its time is also part of RPT_K_EXTEND / RPT_K_SHADOW, which bracket whole queries */
  RPT_K_TREE_SORT = 7,  /* root slab test + queue append (rpt_tree_enter) and the ray sort in front of a traversal;
                           likewise contained in RPT_K_EXTEND / RPT_K_SHADOW */
  RPT_K_COUNT = 8
};

typedef struct RptStats {
  double kernel_ms[RPT_K_COUNT];        /* summed HIP-event time per kernel kind             */
  uint64_t kernel_launches[RPT_K_COUNT];
  uint64_t extend_rays;  /* closest-hit rays traced  (get_closest_hit, renderer.rs:146)      */
  uint64_t shadow_rays;  /* shadow rays the REFERENCE casts for these samples: one per hit and non-ambient
                            light (renderer.rs:191-196)                                       */
  uint64_t shadow_rays_traced; /* of those, the ones actually traversed: a light that can only add exactly zero
                            (bsdf = 0 below an opaque surface, a light sample facing away) needs no ray  */
  uint64_t samples;      /* camera paths started                                             */
  double total_ms;       /* wall time inside rptgpu_render_batch* (host clock)                */
  /* rptgpu_render_batch_reduce since the last reset (ABI v5), HIP events on the library's stream: */
  uint64_t reduce_calls;
  double reduce_render_ms;     /* this rank's own tiles                                         */
  double reduce_collective_ms; /* the gather (or reduce) over the ranks: includes waiting for the slowest one */
  double reduce_copy_ms;       /* root: assembling the frame and its copy to host memory        */
} RptStats;

typedef struct rptgpu_scene rptgpu_scene; /* opaque */

/* ---- library ---- */
int rptgpu_abi_version(void);
const char* rptgpu_strerror(int code);
/* Detail of the last error on this handle (or of the last failed create when h == NULL);
 * the pointer stays valid until the next call on the same handle / thread. */
const char* rptgpu_last_error_detail(const rptgpu_scene* h);
int rptgpu_device_count(int* out_count);

/* ---- scene hand-off: replaces `&Scene` (scene.rs:7-16) behind Renderer::new
 * (renderer.rs:46).  Builds the kd-trees by the reference rule (kdtree.rs:235-345), flattens
 * everything into the device layout and uploads it to `device`.  Unsupported shapes are
 * rejected here, not at render time. */
int rptgpu_scene_create(const RptScene* scene, int device, rptgpu_scene** out);
void rptgpu_scene_destroy(rptgpu_scene* h);

/* ---- the knobs of a scene handle as a struct (ABI v6; SURVEY 5: "kernel tunables via a params struct, not env vars").
 * None of them changes a result — they route work between kernels that compute the same bits (tests: every route
 * against the same fixtures) and bound memory.  rptgpu_scene_options_default() fills in the defaults;
 * rptgpu_scene_create(scene, device, out) is rptgpu_scene_create_opts with them.  The environment variables named
 * beside the fields — the only interface until v5 — remain as OVERRIDES for experiments: read once, inside
 * rptgpu_scene_create[_opts], never afterwards (a handle's behaviour is fixed when it is made).  What is left in the
 * environment only: diagnostics (RPTGPU_PRINT_CREATE / _LAUNCH / _PHASES), the A/B switches RPTGPU_FLAT_TRIS_GLOBAL,
 * RPTGPU_NO_PLANE_TABLE, RPTGPU_PATH_REORDER (the per-depth re-order of the paths in scenes whose trees are walked
 * in-kernel) and RPTGPU_WS_FREE_FRACTION (percent of the free memory a pass may take: 85), the tests' hooks
 * RPTGPU_FAIL_COMM, RPTGPU_PATH_REORDER_MIN and RPTGPU_REC_RATIO, and the launcher's RPTGPU_LOCAL_RANKS /
 * LOCAL_WORLD_SIZE (how many ranks share the host's cores). */
typedef struct RptSceneOptions {
  uint32_t struct_size;           /* sizeof(RptSceneOptions) of the caller's header: lets the struct grow            */
  uint32_t _reserved0;            /* 0                                                                               */
  /* routing */
  uint32_t deep_depth;            /* RPTGPU_DEEP_DEPTH (8): a kd-tree at least this deep gets its own ray queue, sort
                                     and persistent traversal kernel; shallower ones are walked inside the path kernels */
  uint32_t fast_max_depth;        /* RPTGPU_FAST_MAX_DEPTH (32): deeper trees never use the in-kernel traversals        */
  int32_t sort_rays;              /* RPTGPU_SORT_RAYS (-1): 0 never, 1 every deep tree, -1 by the tree's footprint      */
  int32_t rays_in_kernel;         /* RPTGPU_RAYS_IN_KERNEL (0): rptgpu_closest_hit keeps to one kernel for any scene    */
  uint64_t sort_min_bytes;        /* RPTGPU_SORT_MIN_BYTES (8 MiB): nodes + leaf records of a tree whose rays are sorted */
  uint64_t sort_shadow_min_bytes; /* RPTGPU_SORT_SHADOW_MIN_BYTES (8 MiB): ... whose SHADOW rays are sorted too          */
  uint32_t sort_min_rays;         /* RPTGPU_SORT_MIN_RAYS (2^19): queries of fewer rays are not sorted                   */
  int32_t nest_trace;             /* RPTGPU_NEST_TRACE (1): kd-trees of kd-trees in rpt_nest_trace (else rpt_tree_generic) */
  int32_t leaf_boxes;             /* RPTGPU_LEAF_BOXES (1): the conservative f32 box in front of every exact leaf test    */
  int32_t object_filter_min;      /* RPTGPU_OBJECT_FILTER_MIN (5): flat scenes of at least this many objects filter them; 0 = never */
  uint64_t device_build_min;      /* RPTGPU_DEVICE_BUILD_MIN (32768; 4096 when ranks share the host): kd-trees of at least
                                     this many primitives are built on the device; 0 = never                              */
  uint32_t build_threads;         /* RPTGPU_BUILD_THREADS (0 = the usable cores): host threads of the flattening / kd build */
  uint32_t paths_chunk;           /* RPTGPU_PATHS_CHUNK (0 = per launch: 16, 2 for filtered flat scenes at 4+ bounces, fewer
                                     when a lane would get under 24 items): samples per work item of the persistent path kernel       */
  /* memory */
  uint64_t workspace_bytes;       /* RPTGPU_WS_BYTES (240 GiB; and never more than 85 % of the free memory): cap of the
                                     wavefront pipeline's path state                                                          */
  uint64_t lbuf_bytes;            /* RPTGPU_LBUF_BYTES (32 GiB): cap of the per-sample radiance buffer of rpt_paths        */
  uint64_t target_paths;          /* RPTGPU_TARGET_PATHS (0 = what the workspace cap holds, at most 512 Mi): paths per pass */
  /* multi-GPU */
  double comm_timeout_s;          /* RPTGPU_COMM_TIMEOUT_S (300): a batch's exchange is given up after this long           */
  /* routing, added after the first v6 header (struct_size 104): a caller built against that one gets the default */
  int32_t env_park;               /* RPTGPU_ENV_PARK (1): rpt_paths parks the texture lookups of escaped rays per lane and runs
                                     them for the wave together; 0 = each on the spot                                       */
  uint32_t paths_batch;           /* RPTGPU_PATHS_BATCH (0 = per launch: 1/32 of a wave's share of the launch's work items, within
                                     [16, 256]; at most 1024): work items a wave of rpt_paths claims with one atomic on the
                                     work counter                                                                            */
} RptSceneOptions;
/* Fills *out (sizeof(RptSceneOptions) of THIS header, 112 bytes) with the defaults. */
void rptgpu_scene_options_default(RptSceneOptions* out);
/* ABI v7 — the same for a caller whose header may be older: writes exactly struct_size bytes.  struct_size must be a size
 * the struct has had (104: the first v6 header, before env_park; 112); RPTGPU_E_INVALID_ARGUMENT otherwise. */
int rptgpu_scene_options_default_sized(RptSceneOptions* out, uint32_t struct_size);
/* opts == NULL: the defaults.  opts->struct_size must be one of those sizes (the fields a smaller struct lacks take their
 * defaults); RPTGPU_E_INVALID_ARGUMENT otherwise, also when a field — or an RPTGPU_* override of one — is out of range. */
int rptgpu_scene_create_opts(const RptScene* scene, int device, const RptSceneOptions* opts, rptgpu_scene** out);
/* The options a handle runs with, environment overrides applied.  ABI v7: the caller sets out->struct_size to the size of
 * ITS struct before the call (rptgpu_scene_options_default[_sized] leaves it set); exactly that many bytes are written. */
int rptgpu_scene_get_options(const rptgpu_scene* h, RptSceneOptions* out);

/* ---- the hot path: replaces the body of Renderer::sample (renderer.rs:117-129).
 * Writes out_rgb[(y*width+x)*3+c] = mean over `iterations` paths of pixel (x,y), times
 * 2^exposure_value (renderer.rs:141); y = 0 is the top row (renderer.rs:134).  The host then
 * calls Buffer::add_samples(&colors) unchanged (buffer.rs:32-40). */
int rptgpu_render_batch(rptgpu_scene* h, const RptCamera* camera, const RptRenderParams* params,
                        double* out_rgb /* width*height*3, host */);

/* Same, but the result stays in device memory (for an RCCL reduce of the framebuffer by the
 * caller).  `d_out` is a device pointer to width*height*3 elements, f64 if out_is_f32 == 0
 * else f32.  `stream` is a hipStream_t (NULL = the library's own stream); the call returns
 * after the work has been enqueued AND completed on that stream (synchronous). */
int rptgpu_render_batch_device(rptgpu_scene* h, const RptCamera* camera,
                               const RptRenderParams* params, void* d_out, int out_is_f32,
                               void* stream);

/* ---- multi-GPU: one process per GPU, pixel tiles shard across the ranks, ONE collective per batch.
 * The reference's rows are independent (renderer.rs:118-127); rank r renders the 32x8 tiles with
 * tile_id % world == r and the full frame is the SUM of the ranks' frames (every pixel is non-zero on
 * exactly one rank; random numbers are keyed by pixel and sample, so the frame does not depend on the
 * partition).  The library owns the communicator: one RCCL communicator per handle on the handle's
 * device and stream (librccl is opened on first use; without it these calls return RPTGPU_E_COMM).
 *   rank 0:      rptgpu_comm_unique_id(id);  hand the 128 bytes to every rank (any side channel)
 *   every rank:  rptgpu_comm_init(h, rank, world, id);
 *   per batch:   rptgpu_render_batch_reduce(h, camera, params, root, out_rgb32_on_root)
 * rptgpu_render_batch_reduce replaces the body of Renderer::sample on every rank: it renders this
 * rank's tiles (params->tile_*, part_* are overridden: 32x8 tiles, part = rank of world), brings the f32
 * means to `root` over xGMI on the library's stream, and on `root` writes the width*height*3 f32 means to
 * host memory (out_rgb32 may be NULL on other ranks).  Synchronous.  The exchange is a GATHER: every rank
 * sends only the pixels it owns (width*height*3*4 / world bytes, ncclSend / ncclRecv in one group), the
 * root puts them in place — the tiles are disjoint, so nothing is added and the frame is the one a single
 * GPU renders, bit for bit.  RPTGPU_COLLECTIVE=reduce selects ncclReduce(sum) of zero-filled full frames
 * instead (8x the bytes at 8 ranks; the same frame, since every pixel is non-zero on one rank only).
 * With no communicator attached (world = 1) it is a plain single-GPU render into out_rgb32.
 * Failure: the call waits for the stream by polling it together with ncclCommGetAsyncError, at most
 * RPTGPU_COMM_TIMEOUT_S seconds (default 300) per batch.  A rank that fails locally (HIP error, out of
 * memory), sees an asynchronous RCCL error or times out aborts its communicator (ncclCommAbort) and
 * returns RPTGPU_E_COMM / its own error; its peers then run into their time-out or an RCCL error and do
 * the same.  After that every further rptgpu_render_batch_reduce on the handle returns RPTGPU_E_COMM until
 * rptgpu_comm_destroy + rptgpu_comm_init.  The library's stream is drained before the error is returned (nothing is
 * written into out_rgb32 afterwards); should the device not finish within the time-out again, the handle is ABANDONED
 * (ABI v7): its workspace may still be in use by that work, so every later call that would enqueue work on it returns
 * RPTGPU_E_COMM — destroy it.  Argument errors that every rank makes alike (bad params) are
 * returned before anything is enqueued and leave the communicator alone. */
#define RPTGPU_UNIQUE_ID_BYTES 128
int rptgpu_comm_unique_id(uint8_t out_id[RPTGPU_UNIQUE_ID_BYTES]);
int rptgpu_comm_init(rptgpu_scene* h, int rank, int world, const uint8_t id[RPTGPU_UNIQUE_ID_BYTES]);
int rptgpu_comm_destroy(rptgpu_scene* h);
int rptgpu_render_batch_reduce(rptgpu_scene* h, const RptCamera* camera, const RptRenderParams* params,
                               int root, float* out_rgb32 /* width*height*3 f32, host, on root */);
/* Diagnostics: the frame as `world` ranks would produce it, on this one GPU and without a communicator — every
 * rank's part rendered in turn into the root's receive buffer (packed, as ncclSend would deliver it) and placed by
 * the root's pixel lists: the gather of rptgpu_render_batch_reduce minus the wire.  Must equal the frame of a plain
 * render bit for bit (tests; a box with one GPU cannot run RCCL across two ranks). */
int rptgpu_render_batch_emulate_ranks(rptgpu_scene* h, const RptCamera* camera, const RptRenderParams* params,
                                      int world, float* out_rgb32 /* width*height*3 f32, host */);

/* ---- the closest-hit kernel on its own: replaces Renderer::get_closest_hit
 * (renderer.rs:211-220) for a batch of rays (host arrays, n rays, xyz interleaved).
 * out_t = +inf and out_object = -1 on a miss.  A scene with deep trees sends the rays the way a
 * render does (object by object, per-tree queues, sort, persistent traversal); otherwise, or with
 * RPTGPU_RAYS_IN_KERNEL=1, one kernel walks every object per ray.  Same results either way. */
int rptgpu_closest_hit(rptgpu_scene* h, uint64_t n, const double* origins, const double* dirs,
                       uint32_t precision_mode, double* out_t, double* out_normal,
                       int32_t* out_object);

/* ---- host utility: KdTree::new (kdtree.rs:108-119, construct kdtree.rs:235-345) over n
 * axis-aligned boxes (p_min xyz, p_max xyz interleaved: 6 doubles per box).  Returns the
 * flattened tree through malloc'ed arrays the caller releases with rptgpu_free.
 * Node i: split[i], info[i] = axis (0..2) for inner nodes or 3 for a leaf, a[i] = index of
 * the left child (right = a[i]+1) for inner nodes or first entry in refs for a leaf,
 * b[i] = number of refs for a leaf (0 for inner nodes).  Children are visited left first,
 * nodes are numbered in the order they are created by a depth-first construction. */
typedef struct RptKdTree {
  uint64_t num_nodes;
  uint64_t num_refs;
  uint32_t max_depth;
  uint32_t regular; /* 1 if every split plane lies inside its node's cell (rptgpu_kdtree_build only;
                       the device's compact traversal requires it, otherwise the general one runs) */
  double* split;
  uint32_t* info;
  uint32_t* a;
  uint32_t* b;
  uint32_t* refs;
} RptKdTree;
int rptgpu_kdtree_build(const double* boxes, uint64_t n, RptKdTree* out);
/* The same tree — node for node, entry for entry — built on HIP device `device` (events sorted once per axis, one
 * round of scans and stable scatters per tree level).  rptgpu_scene_create uses it by itself for trees of at least
 * RPTGPU_DEVICE_BUILD_MIN primitives (default 32768).  RPTGPU_E_INVALID_ARGUMENT for inputs it does not take (fewer
 * than 16 boxes, non-finite coordinates): build those with rptgpu_kdtree_build. */
int rptgpu_kdtree_build_device(const double* boxes, uint64_t n, int device, RptKdTree* out);
void rptgpu_kdtree_free(RptKdTree* tree);

/* ---- diagnostics: evaluate one function of include/rpt_math.h on the DEVICE for n arguments
 * (host arrays).  fn: 0 exp, 1 log, 2 atan, 3 sin, 4 cos (|x| < 3pi/4), 5 acos, 6 atan2(y, x),
 * 7 y / x computed through the kernels' shared-reciprocal division (must equal IEEE y / x).
 * Lets the parity tests show that the kernels' transcendental functions are bit-identical to
 * the host's. */
int rptgpu_eval_math(rptgpu_scene* h, int fn, uint64_t n, const double* x, const double* y,
                     double* out);

/* ---- device-resident Buffer (reference src/buffer.rs): SURVEY §8f rank 1.
 * Keeps every batch of Renderer::sample on the GPU so that iterative_render (renderer.rs:103-115)
 * needs no host round trip per batch; image() and variance() follow buffer.rs:43-93 exactly:
 * per-pixel batch means summed in insertion order, box filter over (2r+1)^2 neighbours visited
 * x-outer / y-inner, gamma 2.2 + clamp + truncation to u8 (color.rs:18-24). */
typedef struct rptgpu_buffer rptgpu_buffer; /* opaque; belongs to the scene handle it was made from */
int rptgpu_buffer_create(rptgpu_scene* h, uint32_t width, uint32_t height, uint32_t filter_radius,
                         rptgpu_buffer** out);
void rptgpu_buffer_destroy(rptgpu_buffer* b);
/* Renderer::sample(iterations, &mut buffer): render one batch (params->width/height must match the
 * buffer) and Buffer::add_samples it (buffer.rs:32-40), all on the device. */
int rptgpu_buffer_sample(rptgpu_buffer* b, const RptCamera* camera, const RptRenderParams* params);
/* Buffer::image (buffer.rs:43-56): out_rgb8 = height*width*3 bytes (host). */
int rptgpu_buffer_image(rptgpu_buffer* b, uint8_t* out_rgb8);
/* Buffer::variance (buffer.rs:59-73): mean over pixels of the per-pixel sample variance of the batch means. */
int rptgpu_buffer_variance(rptgpu_buffer* b, double* out_variance);
/* number of add_samples calls so far */
int rptgpu_buffer_num_batches(const rptgpu_buffer* b, uint32_t* out);

/* ---- accounting ---- */
int rptgpu_get_stats(const rptgpu_scene* h, RptStats* out);
int rptgpu_reset_stats(rptgpu_scene* h);
/* name of kernel kind k as it appears in a rocprofv3 kernel trace */
const char* rptgpu_kernel_name(int k);

#ifdef __cplusplus
}
#endif
#endif /* RPT_GPU_H */
