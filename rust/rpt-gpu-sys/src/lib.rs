//! `rpt-gpu-sys` — binding of `librptgpu.so`, the MI355X (gfx950) back-end of rpt's one hot path:
//! the body of `Renderer::sample` (rpt `src/renderer.rs:117-129`).
//!
//! Two layers:
//! * [`ffi`]: the `#[repr(C)]` structs and `extern "C"` functions of `include/rpt_gpu.h`, field for
//!   field (checked against the header by `tests/test_rust_layout.py` of the back-end repository);
//! * a small safe layer: [`ShapeDesc`] / [`SceneDesc`] (owned, pointer-free descriptions that rpt's
//!   `Shape::flatten` produces) and [`GpuScene`] (RAII handle).  All `unsafe` is in this crate, so rpt
//!   keeps its `#![forbid(unsafe_code)]` (`src/lib.rs:3`).
//!
//! NOT COMPILED in the back-end's build image (no Rust toolchain there); kept in sync with the header
//! by the layout test.
#![allow(non_camel_case_types)]

use std::ffi::CStr;
use std::fmt;
use std::os::raw::{c_char, c_int, c_void};
use std::sync::Arc;

/// The C ABI of `include/rpt_gpu.h` (ABI version 3).
pub mod ffi {
    use std::os::raw::{c_char, c_int, c_void};

    pub const RPTGPU_ABI_VERSION: c_int = 7;
    pub const RPTGPU_OK: c_int = 0;
    pub const RPTGPU_E_INVALID_ARGUMENT: c_int = -1;
    pub const RPTGPU_E_UNSUPPORTED_SHAPE: c_int = -2;
    pub const RPTGPU_E_NO_DEVICE: c_int = -3;
    pub const RPTGPU_E_HIP: c_int = -4;
    pub const RPTGPU_E_OUT_OF_MEMORY: c_int = -5;
    pub const RPTGPU_E_TREE_TOO_DEEP: c_int = -6;
    pub const RPTGPU_E_UNIMPLEMENTED_SAMPLE: c_int = -7;
    pub const RPTGPU_E_COMM: c_int = -8;

    pub const RPT_SHAPE_SPHERE: i32 = 0;
    pub const RPT_SHAPE_PLANE: i32 = 1;
    pub const RPT_SHAPE_CUBE: i32 = 2;
    pub const RPT_SHAPE_MESH: i32 = 3;
    pub const RPT_SHAPE_GROUP: i32 = 4;
    pub const RPT_SHAPE_MONOMIAL: i32 = 5;
    pub const RPT_LIGHT_POINT: i32 = 0;
    pub const RPT_LIGHT_AMBIENT: i32 = 1;
    pub const RPT_LIGHT_DIRECTIONAL: i32 = 2;
    pub const RPT_LIGHT_OBJECT: i32 = 3;
    pub const RPT_ENV_COLOR: i32 = 0;
    pub const RPT_ENV_HDRI: i32 = 1;
    pub const RPT_PRECISION_F64_STRICT: u32 = 0;
    pub const RPT_FLAG_PROFILE_KERNELS: u32 = 1;
    pub const RPT_FLAG_WAVEFRONT: u32 = 2;
    pub const RPT_FLAG_GENERAL_TRAVERSAL: u32 = 4;
    pub const RPT_FLAG_PERSISTENT: u32 = 8;
    pub const RPT_COLLECTIVE_DEFAULT: u32 = 0;
    pub const RPT_COLLECTIVE_GATHER: u32 = 1;
    pub const RPT_COLLECTIVE_REDUCE: u32 = 2;
    pub const RPT_K_COUNT: usize = 8;
    pub const RPTGPU_UNIQUE_ID_BYTES: usize = 128;

    /// `Material` (rpt src/material.rs:8-26)
    #[repr(C)]
    #[derive(Copy, Clone, Debug)]
    pub struct RptMaterial {
        pub color: [f64; 3],
        pub index: f64,
        pub roughness: f64,
        pub metallic: f64,
        pub emittance: f64,
        pub transparent: i32,
        pub _pad: i32,
    }

    /// `Triangle` (rpt src/shape/mesh.rs:8-22)
    #[repr(C)]
    #[derive(Copy, Clone, Debug)]
    pub struct RptTriangle {
        pub v1: [f64; 3],
        pub v2: [f64; 3],
        pub v3: [f64; 3],
        pub n1: [f64; 3],
        pub n2: [f64; 3],
        pub n3: [f64; 3],
    }

    /// The five precomputed fields of `Transformed<T>` (rpt src/shape.rs:101-108), column-major like nalgebra
    #[repr(C)]
    #[derive(Copy, Clone, Debug)]
    pub struct RptTransform {
        pub transform: [f64; 16],
        pub linear: [f64; 9],
        pub inverse_transform: [f64; 16],
        pub normal_transform: [f64; 9],
        pub scale: f64,
    }

    #[repr(C)]
    #[derive(Copy, Clone)]
    pub struct RptShape {
        pub kind: i32,
        pub transformed: i32,
        pub xf: RptTransform,
        pub plane_normal: [f64; 3],
        pub plane_value: f64,
        pub monomial_height: f64,
        pub monomial_exp: f64,
        pub triangles: *const RptTriangle,
        pub num_triangles: u64,
        pub children: *const RptShape,
        pub num_children: u64,
    }

    /// `Object` (rpt src/object.rs:10-16)
    #[repr(C)]
    #[derive(Copy, Clone)]
    pub struct RptObject {
        pub shape: RptShape,
        pub material: RptMaterial,
    }

    /// `Light` (rpt src/light.rs:7-19)
    #[repr(C)]
    #[derive(Copy, Clone)]
    pub struct RptLight {
        pub kind: i32,
        pub _pad: i32,
        pub color: [f64; 3],
        pub vec: [f64; 3],
        pub object: RptObject,
    }

    /// `Environment` / `Hdri` (rpt src/environment.rs:5-15, 56-62)
    #[repr(C)]
    #[derive(Copy, Clone)]
    pub struct RptEnvironment {
        pub kind: i32,
        pub _pad: i32,
        pub color: [f64; 3],
        pub width: u32,
        pub height: u32,
        pub texels: *const f64,
    }

    /// `Scene` (rpt src/scene.rs:7-16)
    #[repr(C)]
    #[derive(Copy, Clone)]
    pub struct RptScene {
        pub objects: *const RptObject,
        pub num_objects: u64,
        pub lights: *const RptLight,
        pub num_lights: u64,
        pub environment: RptEnvironment,
    }

    /// `Camera` (rpt src/camera.rs:8-26)
    #[repr(C)]
    #[derive(Copy, Clone, Debug)]
    pub struct RptCamera {
        pub eye: [f64; 3],
        pub direction: [f64; 3],
        pub up: [f64; 3],
        pub fov: f64,
        pub aperture: f64,
        pub focal_distance: f64,
    }

    /// What `Renderer` carries into `sample()` (rpt src/renderer.rs:18-42) + seed / partition additions
    #[repr(C)]
    #[derive(Copy, Clone, Debug)]
    pub struct RptRenderParams {
        pub width: u32,
        pub height: u32,
        pub max_bounces: u32,
        pub iterations: u32,
        pub exposure_value: f64,
        pub seed: u64,
        pub sample_index_base: u64,
        pub tile_width: u32,
        pub tile_height: u32,
        pub part_index: u32,
        pub part_count: u32,
        pub precision_mode: u32,
        pub flags: u32,
        pub collective: u32,
        pub _reserved0: u32,
    }

    /// The knobs of a scene handle (ABI v6): `rptgpu_scene_options_default` fills it, `rptgpu_scene_create_opts` takes it.
    #[repr(C)]
    #[derive(Copy, Clone, Debug)]
    pub struct RptSceneOptions {
        pub struct_size: u32,
        pub _reserved0: u32,
        pub deep_depth: u32,
        pub fast_max_depth: u32,
        pub sort_rays: i32,
        pub rays_in_kernel: i32,
        pub sort_min_bytes: u64,
        pub sort_shadow_min_bytes: u64,
        pub sort_min_rays: u32,
        pub nest_trace: i32,
        pub leaf_boxes: i32,
        pub object_filter_min: i32,
        pub device_build_min: u64,
        pub build_threads: u32,
        pub paths_chunk: u32,
        pub workspace_bytes: u64,
        pub lbuf_bytes: u64,
        pub target_paths: u64,
        pub comm_timeout_s: f64,
        pub env_park: i32,
        pub paths_batch: u32,
    }

    #[repr(C)]
    #[derive(Copy, Clone, Debug)]
    pub struct RptStats {
        pub kernel_ms: [f64; 8],
        pub kernel_launches: [u64; 8],
        pub extend_rays: u64,
        pub shadow_rays: u64,
        pub shadow_rays_traced: u64,
        pub samples: u64,
        pub total_ms: f64,
        pub reduce_calls: u64,
        pub reduce_render_ms: f64,
        pub reduce_collective_ms: f64,
        pub reduce_copy_ms: f64,
    }

    #[repr(C)]
    pub struct RptKdTree {
        pub num_nodes: u64,
        pub num_refs: u64,
        pub max_depth: u32,
        pub regular: u32,
        pub split: *mut f64,
        pub info: *mut u32,
        pub a: *mut u32,
        pub b: *mut u32,
        pub refs: *mut u32,
    }

    /// opaque `rptgpu_scene`
    #[repr(C)]
    pub struct rptgpu_scene {
        _private: [u8; 0],
    }
    /// opaque `rptgpu_buffer`
    #[repr(C)]
    pub struct rptgpu_buffer {
        _private: [u8; 0],
    }

    extern "C" {
        pub fn rptgpu_abi_version() -> c_int;
        pub fn rptgpu_strerror(code: c_int) -> *const c_char;
        pub fn rptgpu_last_error_detail(h: *const rptgpu_scene) -> *const c_char;
        pub fn rptgpu_device_count(out_count: *mut c_int) -> c_int;
        pub fn rptgpu_scene_create(scene: *const RptScene, device: c_int, out: *mut *mut rptgpu_scene) -> c_int;
        pub fn rptgpu_scene_destroy(h: *mut rptgpu_scene);
        pub fn rptgpu_scene_options_default(out: *mut RptSceneOptions);
        pub fn rptgpu_scene_options_default_sized(out: *mut RptSceneOptions, struct_size: u32) -> c_int;
        pub fn rptgpu_scene_create_opts(scene: *const RptScene, device: c_int, opts: *const RptSceneOptions, out: *mut *mut rptgpu_scene) -> c_int;
        pub fn rptgpu_scene_get_options(h: *const rptgpu_scene, out: *mut RptSceneOptions) -> c_int;
        pub fn rptgpu_render_batch(h: *mut rptgpu_scene, camera: *const RptCamera, params: *const RptRenderParams, out_rgb: *mut f64) -> c_int;
        pub fn rptgpu_render_batch_device(h: *mut rptgpu_scene, camera: *const RptCamera, params: *const RptRenderParams, d_out: *mut c_void, out_is_f32: c_int, stream: *mut c_void) -> c_int;
        pub fn rptgpu_comm_unique_id(out_id: *mut u8) -> c_int;
        pub fn rptgpu_comm_init(h: *mut rptgpu_scene, rank: c_int, world: c_int, id: *const u8) -> c_int;
        pub fn rptgpu_comm_destroy(h: *mut rptgpu_scene) -> c_int;
        pub fn rptgpu_render_batch_reduce(h: *mut rptgpu_scene, camera: *const RptCamera, params: *const RptRenderParams, root: c_int, out_rgb32: *mut f32) -> c_int;
        pub fn rptgpu_render_batch_emulate_ranks(h: *mut rptgpu_scene, camera: *const RptCamera, params: *const RptRenderParams, world: c_int, out_rgb32: *mut f32) -> c_int;
        pub fn rptgpu_closest_hit(h: *mut rptgpu_scene, n: u64, origins: *const f64, dirs: *const f64, precision_mode: u32, out_t: *mut f64, out_normal: *mut f64, out_object: *mut i32) -> c_int;
        pub fn rptgpu_kdtree_build(boxes: *const f64, n: u64, out: *mut RptKdTree) -> c_int;
        pub fn rptgpu_kdtree_build_device(boxes: *const f64, n: u64, device: c_int, out: *mut RptKdTree) -> c_int;
        pub fn rptgpu_kdtree_free(tree: *mut RptKdTree);
        pub fn rptgpu_eval_math(h: *mut rptgpu_scene, func: c_int, n: u64, x: *const f64, y: *const f64, out: *mut f64) -> c_int;
        pub fn rptgpu_buffer_create(h: *mut rptgpu_scene, width: u32, height: u32, filter_radius: u32, out: *mut *mut rptgpu_buffer) -> c_int;
        pub fn rptgpu_buffer_destroy(b: *mut rptgpu_buffer);
        pub fn rptgpu_buffer_sample(b: *mut rptgpu_buffer, camera: *const RptCamera, params: *const RptRenderParams) -> c_int;
        pub fn rptgpu_buffer_image(b: *mut rptgpu_buffer, out_rgb8: *mut u8) -> c_int;
        pub fn rptgpu_buffer_variance(b: *mut rptgpu_buffer, out_variance: *mut f64) -> c_int;
        pub fn rptgpu_buffer_num_batches(b: *const rptgpu_buffer, out: *mut u32) -> c_int;
        pub fn rptgpu_get_stats(h: *const rptgpu_scene, out: *mut RptStats) -> c_int;
        pub fn rptgpu_reset_stats(h: *mut rptgpu_scene) -> c_int;
        pub fn rptgpu_kernel_name(k: c_int) -> *const c_char;
    }
}

pub use ffi::{RptCamera, RptMaterial, RptRenderParams, RptSceneOptions, RptStats, RptTransform, RptTriangle};

// ------------------------------------------------------------------------------------------------
// Safe layer
// ------------------------------------------------------------------------------------------------

/// An error code of the library plus its detail string.
#[derive(Debug, Clone)]
pub struct GpuError {
    pub code: i32,
    pub message: String,
}

impl fmt::Display for GpuError {
    fn fmt(&self, f: &mut fmt::Formatter<'_>) -> fmt::Result {
        write!(f, "rptgpu error {}: {}", self.code, self.message)
    }
}
impl std::error::Error for GpuError {}

fn cstr(p: *const c_char) -> String {
    if p.is_null() {
        String::new()
    } else {
        // SAFETY: the library returns NUL-terminated static or handle-owned strings
        unsafe { CStr::from_ptr(p) }.to_string_lossy().into_owned()
    }
}

fn check(code: c_int, h: *const ffi::rptgpu_scene) -> Result<(), GpuError> {
    if code == ffi::RPTGPU_OK {
        return Ok(());
    }
    // SAFETY: plain queries; `h` may be null (detail of the last failed create on this thread)
    let (what, detail) = unsafe { (cstr(ffi::rptgpu_strerror(code)), cstr(ffi::rptgpu_last_error_detail(h))) };
    Err(GpuError { code, message: if detail.is_empty() { what } else { format!("{} — {}", what, detail) } })
}

/// The closed shape set of the device, as an owned, pointer-free tree.  This is what rpt's
/// `Shape::flatten` returns (`rust/rpt.patch`): `Sphere` (sphere.rs:9), `Plane` (plane.rs:7-13), `Cube`
/// (cube.rs:8), `Mesh` = `KdTree<Triangle>` (mesh.rs:102), `Group` = `KdTree<Box<dyn Bounded>>`
/// (kdtree.rs:14-24), `Monomial` (monomial_surface.rs:12-18) and the `Transformed<T>` wrapper
/// (shape.rs:101-108).  A mesh is shared by `Arc`: instances of one `Arc<Mesh>` share one device tree.
#[derive(Clone)]
pub enum ShapeDesc {
    Sphere,
    Plane { normal: [f64; 3], value: f64 },
    Cube,
    Mesh(Arc<Vec<RptTriangle>>),
    Group(Vec<ShapeDesc>),
    Monomial { height: f64, exp: f64 },
    Transformed { inner: Box<ShapeDesc>, xf: RptTransform },
}

/// `Light` (light.rs:7-19)
#[derive(Clone)]
pub enum LightDesc {
    Point { color: [f64; 3], location: [f64; 3] },
    Ambient { color: [f64; 3] },
    Directional { color: [f64; 3], direction: [f64; 3] },
    Object { shape: ShapeDesc, material: RptMaterial },
}

/// `Environment` (environment.rs:56-62)
#[derive(Clone)]
pub enum EnvDesc {
    Color([f64; 3]),
    /// width, height, width*height*3 texels, row-major (environment.rs:5-15)
    Hdri { width: u32, height: u32, texels: Arc<Vec<f64>> },
}

/// `Scene` (scene.rs:7-16) in flattened form
#[derive(Clone)]
pub struct SceneDesc {
    pub objects: Vec<(ShapeDesc, RptMaterial)>,
    pub lights: Vec<LightDesc>,
    pub environment: EnvDesc,
}

const IDENTITY_XF: RptTransform = RptTransform {
    transform: [0.0; 16],
    linear: [0.0; 9],
    inverse_transform: [0.0; 16],
    normal_transform: [0.0; 9],
    scale: 0.0,
};

/// Keeps the child arrays of GROUP shapes alive while the C description points into them.
#[derive(Default)]
struct Lowering {
    child_arrays: Vec<Box<[ffi::RptShape]>>,
}

impl Lowering {
    fn shape(&mut self, s: &ShapeDesc) -> ffi::RptShape {
        let mut out = ffi::RptShape {
            kind: 0,
            transformed: 0,
            xf: IDENTITY_XF,
            plane_normal: [0.0; 3],
            plane_value: 0.0,
            monomial_height: 0.0,
            monomial_exp: 0.0,
            triangles: std::ptr::null(),
            num_triangles: 0,
            children: std::ptr::null(),
            num_children: 0,
        };
        match s {
            ShapeDesc::Sphere => out.kind = ffi::RPT_SHAPE_SPHERE,
            ShapeDesc::Cube => out.kind = ffi::RPT_SHAPE_CUBE,
            ShapeDesc::Plane { normal, value } => {
                out.kind = ffi::RPT_SHAPE_PLANE;
                out.plane_normal = *normal;
                out.plane_value = *value;
            }
            ShapeDesc::Monomial { height, exp } => {
                out.kind = ffi::RPT_SHAPE_MONOMIAL;
                out.monomial_height = *height;
                out.monomial_exp = *exp;
            }
            ShapeDesc::Mesh(tris) => {
                out.kind = ffi::RPT_SHAPE_MESH;
                out.triangles = tris.as_ptr(); // the Arc is kept alive by the SceneDesc for the duration of the call
                out.num_triangles = tris.len() as u64;
            }
            ShapeDesc::Group(kids) => {
                out.kind = ffi::RPT_SHAPE_GROUP;
                let lowered: Vec<ffi::RptShape> = kids.iter().map(|k| self.shape(k)).collect();
                let boxed = lowered.into_boxed_slice();
                out.children = boxed.as_ptr();
                out.num_children = boxed.len() as u64;
                self.child_arrays.push(boxed);
            }
            ShapeDesc::Transformed { inner, xf } => {
                // Transformed<Transformed<T>> does not occur: rpt's builders compose the matrices
                // (shape.rs:202-284), so `inner` is a bare shape
                out = self.shape(inner);
                out.transformed = 1;
                out.xf = *xf;
            }
        }
        out
    }
}

/// Owner of one `rptgpu_scene*`: the device-resident flattened scene with its kd-trees.
pub struct GpuScene {
    h: *mut ffi::rptgpu_scene,
}

// The handle is not re-entrant (one render in flight per handle), which `&mut self` on the render
// methods enforces; it may move between threads.
unsafe impl Send for GpuScene {}

impl GpuScene {
    /// Number of HIP devices the library sees.
    pub fn device_count() -> Result<i32, GpuError> {
        let mut n: c_int = 0;
        // SAFETY: out-pointer to a local
        check(unsafe { ffi::rptgpu_device_count(&mut n) }, std::ptr::null())?;
        Ok(n)
    }

    /// The back-end's knobs with the library's defaults (`rptgpu_scene_options_default`): change fields, then
    /// `GpuScene::with_options`.  None of them changes a result.
    pub fn default_options() -> RptSceneOptions {
        let mut o = std::mem::MaybeUninit::<RptSceneOptions>::zeroed();
        // SAFETY: the library writes exactly the size it is told (ABI v7), i.e. THIS crate's struct even when the
        // library's own has grown since; an all-zero RptSceneOptions is a valid value should the size be unknown to it
        unsafe {
            let _ = ffi::rptgpu_scene_options_default_sized(o.as_mut_ptr(), std::mem::size_of::<RptSceneOptions>() as u32);
            o.assume_init()
        }
    }

    /// `rptgpu_scene_create`: kd construction by the reference rule (kdtree.rs:235-345), flattening, upload.
    pub fn new(scene: &SceneDesc, device: i32) -> Result<Self, GpuError> {
        Self::create(scene, device, None)
    }

    /// `rptgpu_scene_create_opts`: the same with explicit options (ABI v6).
    pub fn with_options(scene: &SceneDesc, device: i32, options: &RptSceneOptions) -> Result<Self, GpuError> {
        Self::create(scene, device, Some(options))
    }

    /// The options the handle runs with: defaults, the caller's, `RPTGPU_*` environment overrides.
    pub fn options(&self) -> Result<RptSceneOptions, GpuError> {
        let mut o = Self::default_options();
        // SAFETY: a live handle; `o.struct_size` (set by default_options) tells the library how many bytes it may write
        check(unsafe { ffi::rptgpu_scene_get_options(self.h, &mut o) }, self.h)?;
        Ok(o)
    }

    fn create(scene: &SceneDesc, device: i32, options: Option<&RptSceneOptions>) -> Result<Self, GpuError> {
        // SAFETY (whole function): every pointer placed in the C description points into `scene`, `low` or the
        // local vectors below, all of which outlive the call; the library copies what it needs.
        if unsafe { ffi::rptgpu_abi_version() } != ffi::RPTGPU_ABI_VERSION {
            return Err(GpuError { code: ffi::RPTGPU_E_INVALID_ARGUMENT, message: "librptgpu.so ABI version mismatch".into() });
        }
        let mut low = Lowering::default();
        let objects: Vec<ffi::RptObject> =
            scene.objects.iter().map(|(s, m)| ffi::RptObject { shape: low.shape(s), material: *m }).collect();
        let no_material = RptMaterial { color: [0.0; 3], index: 0.0, roughness: 0.0, metallic: 0.0, emittance: 0.0, transparent: 0, _pad: 0 };
        let lights: Vec<ffi::RptLight> = scene
            .lights
            .iter()
            .map(|l| {
                let mut out = ffi::RptLight {
                    kind: 0,
                    _pad: 0,
                    color: [0.0; 3],
                    vec: [0.0; 3],
                    object: ffi::RptObject { shape: low.shape(&ShapeDesc::Sphere), material: no_material },
                };
                match l {
                    LightDesc::Point { color, location } => {
                        out.kind = ffi::RPT_LIGHT_POINT;
                        out.color = *color;
                        out.vec = *location;
                    }
                    LightDesc::Ambient { color } => {
                        out.kind = ffi::RPT_LIGHT_AMBIENT;
                        out.color = *color;
                    }
                    LightDesc::Directional { color, direction } => {
                        out.kind = ffi::RPT_LIGHT_DIRECTIONAL;
                        out.color = *color;
                        out.vec = *direction;
                    }
                    LightDesc::Object { shape, material } => {
                        out.kind = ffi::RPT_LIGHT_OBJECT;
                        out.object = ffi::RptObject { shape: low.shape(shape), material: *material };
                    }
                }
                out
            })
            .collect();
        let environment = match &scene.environment {
            EnvDesc::Color(c) => ffi::RptEnvironment { kind: ffi::RPT_ENV_COLOR, _pad: 0, color: *c, width: 0, height: 0, texels: std::ptr::null() },
            EnvDesc::Hdri { width, height, texels } => {
                assert_eq!(texels.len(), (*width as usize) * (*height as usize) * 3);
                ffi::RptEnvironment { kind: ffi::RPT_ENV_HDRI, _pad: 0, color: [0.0; 3], width: *width, height: *height, texels: texels.as_ptr() }
            }
        };
        let desc = ffi::RptScene {
            objects: objects.as_ptr(),
            num_objects: objects.len() as u64,
            lights: lights.as_ptr(),
            num_lights: lights.len() as u64,
            environment,
        };
        let mut h: *mut ffi::rptgpu_scene = std::ptr::null_mut();
        let opts = options.map_or(std::ptr::null(), |o| o as *const RptSceneOptions);
        check(unsafe { ffi::rptgpu_scene_create_opts(&desc, device as c_int, opts, &mut h) }, std::ptr::null())?;
        drop(low);
        Ok(GpuScene { h })
    }

    /// The body of `Renderer::sample`: `out_rgb[(y*width+x)*3+c]` = mean over `params.iterations` paths times
    /// 2^EV (renderer.rs:131-142); the caller then calls `Buffer::add_samples` unchanged (buffer.rs:32-40).
    pub fn render_batch(&mut self, camera: &RptCamera, params: &RptRenderParams, out_rgb: &mut [f64]) -> Result<(), GpuError> {
        assert_eq!(out_rgb.len(), params.width as usize * params.height as usize * 3, "Invalid sample dimension");
        // SAFETY: the slice has exactly the size the library writes
        check(unsafe { ffi::rptgpu_render_batch(self.h, camera, params, out_rgb.as_mut_ptr()) }, self.h)
    }

    /// Multi-GPU form (one process per GPU): renders this rank's tiles, gathers the owned pixels on `root` inside the
    /// library (RCCL send / receive; `params.collective = RPT_COLLECTIVE_REDUCE` for an `ncclReduce`) and fills `out_rgb32` on the root rank.
    pub fn render_batch_reduce(&mut self, camera: &RptCamera, params: &RptRenderParams, root: i32, out_rgb32: &mut [f32]) -> Result<(), GpuError> {
        assert_eq!(out_rgb32.len(), params.width as usize * params.height as usize * 3);
        // SAFETY: as above
        check(unsafe { ffi::rptgpu_render_batch_reduce(self.h, camera, params, root as c_int, out_rgb32.as_mut_ptr()) }, self.h)
    }

    /// 128 bytes rank 0 hands to every rank (any side channel) for [`GpuScene::comm_init`].
    pub fn comm_unique_id() -> Result<[u8; ffi::RPTGPU_UNIQUE_ID_BYTES], GpuError> {
        let mut id = [0u8; ffi::RPTGPU_UNIQUE_ID_BYTES];
        // SAFETY: 128-byte out buffer
        check(unsafe { ffi::rptgpu_comm_unique_id(id.as_mut_ptr()) }, std::ptr::null())?;
        Ok(id)
    }

    pub fn comm_init(&mut self, rank: i32, world: i32, id: &[u8; ffi::RPTGPU_UNIQUE_ID_BYTES]) -> Result<(), GpuError> {
        // SAFETY: 128-byte in buffer
        check(unsafe { ffi::rptgpu_comm_init(self.h, rank as c_int, world as c_int, id.as_ptr()) }, self.h)
    }

    /// `Renderer::get_closest_hit` (renderer.rs:211-220) for a batch of rays: (t, normal, object index or -1).
    pub fn closest_hit(&mut self, origins: &[[f64; 3]], dirs: &[[f64; 3]]) -> Result<(Vec<f64>, Vec<[f64; 3]>, Vec<i32>), GpuError> {
        assert_eq!(origins.len(), dirs.len());
        let n = origins.len();
        let (mut t, mut nrm, mut obj) = (vec![0.0f64; n], vec![[0.0f64; 3]; n], vec![0i32; n]);
        // SAFETY: [[f64; 3]] is 3 contiguous f64 per element; all buffers hold n elements
        check(
            unsafe {
                ffi::rptgpu_closest_hit(self.h, n as u64, origins.as_ptr() as *const f64, dirs.as_ptr() as *const f64, 0,
                                        t.as_mut_ptr(), nrm.as_mut_ptr() as *mut f64, obj.as_mut_ptr())
            },
            self.h,
        )?;
        Ok((t, nrm, obj))
    }

    pub fn stats(&self) -> Result<RptStats, GpuError> {
        let mut s = RptStats { kernel_ms: [0.0; 8], kernel_launches: [0; 8], extend_rays: 0, shadow_rays: 0, shadow_rays_traced: 0, samples: 0, total_ms: 0.0, reduce_calls: 0, reduce_render_ms: 0.0, reduce_collective_ms: 0.0, reduce_copy_ms: 0.0 };
        // SAFETY: out-pointer to a local
        check(unsafe { ffi::rptgpu_get_stats(self.h, &mut s) }, self.h)?;
        Ok(s)
    }

    /// Raw handle, for the entry points this safe layer does not wrap (`rptgpu_buffer_*`, `rptgpu_render_batch_device`).
    pub fn as_ptr(&mut self) -> *mut c_void {
        self.h as *mut c_void
    }
}

impl Drop for GpuScene {
    fn drop(&mut self) {
        // SAFETY: the handle came from rptgpu_scene_create and is destroyed exactly once
        unsafe { ffi::rptgpu_scene_destroy(self.h) }
    }
}

#[cfg(test)]
mod layout_tests {
    //! `cargo test` in this crate re-checks what tests/test_rust_layout.py of the back-end repository checks from the
    //! outside: the sizes of the `#[repr(C)]` structs are those of the C structs (x86-64 / LP64; values from gcc via
    //! tests/test_abi.py), and the library this crate links to speaks the same ABI version.
    use super::ffi::*;
    use std::mem::size_of;

    #[test]
    fn struct_sizes_match_rpt_gpu_h() {
        assert_eq!(size_of::<RptMaterial>(), 64);
        assert_eq!(size_of::<RptTriangle>(), 144);
        assert_eq!(size_of::<RptTransform>(), 408);
        assert_eq!(size_of::<RptShape>(), 496);
        assert_eq!(size_of::<RptObject>(), 560);
        assert_eq!(size_of::<RptLight>(), 616);
        assert_eq!(size_of::<RptEnvironment>(), 48);
        assert_eq!(size_of::<RptScene>(), 80);
        assert_eq!(size_of::<RptCamera>(), 96);
        assert_eq!(size_of::<RptRenderParams>(), 72);
        assert_eq!(size_of::<RptSceneOptions>(), 112);
        assert_eq!(size_of::<RptStats>(), 200);
        assert_eq!(size_of::<RptKdTree>(), 64);
    }

    #[test]
    fn abi_version_matches_the_library() {
        // SAFETY: plain query
        assert_eq!(unsafe { rptgpu_abi_version() }, RPTGPU_ABI_VERSION);
    }
}
