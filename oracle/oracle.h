/*
 * oracle.h — C entry points of the CPU oracle (TEST INFRASTRUCTURE, not product code).
 *
 * The oracle is a plain C++17, IEEE-f64, no-FMA restatement of rpt's CPU path
 * (reference src/renderer.rs, kdtree.rs, shape*.rs, material.rs, light.rs, camera.rs,
 * environment.rs, buffer.rs, color.rs).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  It consumes the boundary's POD scene description
 * (include/rpt_gpu.h) so that the same inputs can be handed to the HIP path.
 *
 * PARITY STATUS: "parity unpinned" against the reference itself.  The Rust reference cannot
 * be built here (no cargo/rustc, no vendored crates), its renderer seeds from OS entropy
 * (renderer.rs:121) and its only test next to this path is `colors_work` (color.rs:31-38),
 * which IS reproduced (tests/test_oracle_color.py).  Everything else is pinned by closed-form
 * known-answer tests derived from the cited lines (tests/test_oracle_kat.py).
 */
#ifndef RPT_ORACLE_H
#define RPT_ORACLE_H

#include <stdint.h>
#include "../include/rpt_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* visit counters of the reference algorithm; they define the algorithmic-bytes figure
 * (SURVEY.md §8d).  All counts are totals over the call. */
typedef struct OracleCounters {
  uint64_t samples;      /* camera paths                                                   */
  uint64_t segments;     /* trace_ray activations (renderer.rs:145)                        */
  uint64_t closest_rays; /* get_closest_hit calls from trace_ray (renderer.rs:146)         */
  uint64_t shadow_rays;  /* get_closest_hit calls from sample_lights (renderer.rs:191)     */
  uint64_t hits;         /* trace_ray activations that hit an object                       */
  uint64_t misses;       /* ... that returned the environment                              */
  uint64_t n_inst;       /* Transformed::intersect calls (shape.rs:128)                    */
  uint64_t n_root;       /* KdTree::intersect root slab tests (kdtree.rs:130)              */
  uint64_t n_inner;      /* intersect_subtree calls on split nodes (kdtree.rs:172-204)     */
  uint64_t n_leaf;       /* intersect_subtree calls on leaves (kdtree.rs:162)              */
  uint64_t n_ref;        /* leaf entries visited (kdtree.rs:165)                           */
  uint64_t n_tri;        /* Triangle::intersect calls (mesh.rs:49)                         */
  uint64_t n_sphere;     /* Sphere::intersect calls (sphere.rs:13)                         */
  uint64_t n_plane;      /* Plane::intersect calls (plane.rs:17)                           */
  uint64_t n_cube;       /* Cube::intersect calls (cube.rs:20)                             */
  uint64_t rng_draws;    /* next_u64 calls                                                 */
  /* the same geometry counters for work done on behalf of SHADOW rays (renderer.rs:191-196);
   * the unsuffixed ones above count closest-hit rays from trace_ray only */
  uint64_t n_inst_sh, n_root_sh, n_inner_sh, n_leaf_sh, n_ref_sh, n_tri_sh, n_sphere_sh, n_plane_sh,
      n_cube_sh;
  uint64_t n_monomial, n_monomial_sh; /* MonomialSurface::intersect calls (monomial_surface.rs:21) */
} OracleCounters;

typedef struct oracle_scene oracle_scene;

/* Scene::from the boundary description.  Builds KdTree::new (kdtree.rs:108-119) itself. */
int oracle_scene_create(const RptScene* scene, oracle_scene** out);
void oracle_scene_destroy(oracle_scene* s);

/* Renderer::sample (renderer.rs:117-129): out_rgb[(y*W+x)*3+c]; rows are claimed
 * dynamically by `threads` std::threads (one task per row, as rayon does).  Honours the
 * same tile partition fields as the product (unrendered pixels are written as 0). */
int oracle_render(const oracle_scene* s, const RptCamera* cam, const RptRenderParams* p,
                  int threads, double* out_rgb, OracleCounters* counters /* may be NULL */);

/* One path: L(ray,0) for pixel (x,y), sample s — with its per-depth records:
 * rec[k*8 + 0..2] = A_k, +3..5 = f, +6 = 1/pdf, +7 = |wi.n|; returns number of records. */
int oracle_trace_sample(const oracle_scene* s, const RptCamera* cam, const RptRenderParams* p,
                        uint32_t x, uint32_t y, uint64_t sample, double* out_rgb,
                        double* rec /* (max_bounces+1)*8 or NULL */, int* out_nrec);

/* Renderer::get_closest_hit (renderer.rs:211-220) for n rays. */
int oracle_closest_hit(const oracle_scene* s, uint64_t n, const double* origins,
                       const double* dirs, double* out_t, double* out_normal,
                       int32_t* out_object, OracleCounters* counters);

/* camera rays: Camera::cast_ray through Renderer::get_color's pixel mapping
 * (renderer.rs:131-139) for pixel (x,y), sample s: out = origin xyz, dir xyz */
int oracle_camera_ray(const RptCamera* cam, const RptRenderParams* p, uint32_t x, uint32_t y,
                      uint64_t sample, double* out6);

/* Material::bsdf (material.rs:125-210) */
void oracle_bsdf(const RptMaterial* m, const double* n, const double* wo, const double* wi,
                 double* out3);
/* Material::sample_f (material.rs:224-313) with the Philox stream (seed,pixel,sample) starting
 * at draw index *draw; returns 1 for Some, 0 for None; advances *draw. */
int oracle_sample_f(const RptMaterial* m, const double* n, const double* wo, uint64_t seed,
                    uint32_t pixel, uint64_t sample, uint32_t* draw, double* out_wi,
                    double* out_pdf);
/* Light::illuminate (light.rs:23-47): out = intensity xyz, wi xyz, dist */
int oracle_illuminate(const RptLight* light, const double* pos, uint64_t seed, uint32_t pixel,
                      uint64_t sample, uint32_t* draw, double* out7);
/* Environment::get_color (environment.rs:72-77) */
void oracle_env_color(const RptEnvironment* env, const double* dir, double* out3);

/* single-shape intersect (Shape::intersect, shape.rs:21) with an explicit record.time;
 * returns 1 if the record was updated */
int oracle_shape_intersect(const RptShape* shape, const double* origin, const double* dir,
                           double t_min, double* inout_time, double* out_normal);
/* Shape::sample (shape.rs:24): out = point xyz, normal xyz, pdf */
int oracle_shape_sample(const RptShape* shape, const double* target, uint64_t seed,
                        uint32_t pixel, uint64_t sample, uint32_t* draw, double* out7);

/* MonomialSurface::closest_point (monomial_surface.rs:126-152, steps = 100) and
 * closest_point_precise (:154-181, steps = 10000) — not on the render path; here so that the
 * reference's own unit test `monomial_closest_point_works` (:192-240) can be replayed. */
void oracle_monomial_closest_point(double height, const double* point, int steps, double* out3);

/* BoundingBox::intersect (kdtree.rs:54-68): box = pmin xyz, pmax xyz */
void oracle_bbox_intersect(const double* box6, const double* origin, const double* dir,
                           double* out_min, double* out_max);

/* KdTree::new over boxes (kdtree.rs:108-119,235-345); same flattened form as
 * rptgpu_kdtree_build so the two can be compared node by node. */
int oracle_kdtree_build(const double* boxes, uint64_t n, RptKdTree* out);
void oracle_kdtree_free(RptKdTree* t);

/* RNG: Philox4x32-10 block and the rand / rand_distr distribution algorithms */
void oracle_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
uint64_t oracle_rng_u64(uint64_t seed, uint32_t pixel, uint64_t sample, uint32_t draw);
/* kind: 0 gen::<f64>, 1 gen_range(lo..hi), 2 gen_bool(p=lo), 3 Uniform(0..n=lo),
 *       4 UnitDisc, 5 UnitCircle ; writes 1 or 2 doubles, advances *draw */
int oracle_rng_sample(int kind, double lo, double hi, uint64_t seed, uint32_t pixel,
                      uint64_t sample, uint32_t* draw, double* out2);

/* include/rpt_math.h evaluated on the host: fn 0 exp, 1 log, 2 atan, 3 sin, 4 cos (|x| < 3pi/4),
 * 5 acos, 6 atan2(y, x) */
void oracle_math_eval(int fn, uint64_t n, const double* x, const double* y, double* out);

/* color.rs:10-24 */
void oracle_hex_color(uint32_t x, double* out3);
void oracle_color_bytes(const double* color3, uint8_t* out3);
/* Buffer (buffer.rs): batches = nb arrays of W*H*3 doubles (one add_samples call each) */
void oracle_buffer_image(uint32_t w, uint32_t h, uint32_t radius, uint32_t nb,
                         const double* const* batches, uint8_t* out_rgb8);
double oracle_buffer_variance(uint32_t w, uint32_t h, uint32_t nb, const double* const* batches);

#ifdef __cplusplus
}
#endif
#endif
