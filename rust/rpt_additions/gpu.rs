//! Feature `gpu`: `Renderer::sample` through `librptgpu.so`, the MI355X (gfx950) back-end.
//!
//! Everything here is safe code: `Shape::flatten` turns the scene into `rpt_gpu_sys::ShapeDesc` values (owned,
//! pointer-free), and the `rpt-gpu-sys` crate lowers those to the C ABI of `include/rpt_gpu.h`.  If any object or
//! light shape is outside the back-end's closed set (`flatten` returns `None`), `GpuBackend::new` returns `None`
//! and the renderer keeps its rayon path.
use std::collections::HashMap;
use std::sync::{Arc, Mutex};

use rpt_gpu_sys::{EnvDesc, GpuScene, LightDesc, RptCamera, RptMaterial, RptRenderParams, RptTriangle, SceneDesc, ShapeDesc};

use crate::camera::Camera;
use crate::color::Color;
use crate::environment::Environment;
use crate::light::Light;
use crate::material::Material;
use crate::renderer::Renderer;
use crate::scene::Scene;
use crate::shape::Triangle;

/// Scratch state of one flattening pass: triangle arrays already converted, keyed by the address of the mesh's
/// triangle slice, so that every instance of one `Arc<Mesh>` refers to one array (and one device kd-tree)
#[derive(Default)]
pub struct FlatArena {
    meshes: HashMap<usize, Arc<Vec<RptTriangle>>>,
}

fn v3(v: &glm::DVec3) -> [f64; 3] {
    [v.x, v.y, v.z]
}

impl FlatArena {
    /// The triangles of a `KdTree<Triangle>` in the ABI's layout (v1 v2 v3 n1 n2 n3, mesh.rs:8-22)
    pub fn mesh(&mut self, triangles: &[Triangle]) -> Arc<Vec<RptTriangle>> {
        let key = triangles.as_ptr() as usize;
        self.meshes
            .entry(key)
            .or_insert_with(|| {
                Arc::new(
                    triangles
                        .iter()
                        .map(|t| RptTriangle { v1: v3(&t.v1), v2: v3(&t.v2), v3: v3(&t.v3), n1: v3(&t.n1), n2: v3(&t.n2), n3: v3(&t.n3) })
                        .collect(),
                )
            })
            .clone()
    }
}

fn material(m: &Material) -> RptMaterial {
    RptMaterial {
        color: v3(&m.color),
        index: m.index,
        roughness: m.roughness,
        metallic: m.metallic,
        emittance: m.emittance,
        transparent: m.transparent as i32,
        _pad: 0,
    }
}

fn camera(c: &Camera) -> RptCamera {
    RptCamera { eye: v3(&c.eye), direction: v3(&c.direction), up: v3(&c.up), fov: c.fov, aperture: c.aperture, focal_distance: c.focal_distance }
}

/// `Scene` (scene.rs:7-16) -> `SceneDesc`; `None` if a shape is outside the device's closed set
pub fn flatten_scene(scene: &Scene) -> Option<SceneDesc> {
    let mut arena = FlatArena::default();
    let mut objects = Vec::with_capacity(scene.objects.len());
    for o in &scene.objects {
        objects.push((o.shape.flatten(&mut arena)?, material(&o.material)));
    }
    let mut lights = Vec::with_capacity(scene.lights.len());
    for l in &scene.lights {
        lights.push(match l {
            Light::Point(color, location) => LightDesc::Point { color: v3(color), location: v3(location) },
            Light::Ambient(color) => LightDesc::Ambient { color: v3(color) },
            Light::Directional(color, direction) => LightDesc::Directional { color: v3(color), direction: v3(direction) },
            Light::Object(o) => LightDesc::Object { shape: o.shape.flatten(&mut arena)?, material: material(&o.material) },
        });
    }
    let environment = match &scene.environment {
        Environment::Color(c) => EnvDesc::Color(v3(c)),
        Environment::Hdri(h) => {
            let (width, height, buf) = h.raw();
            EnvDesc::Hdri { width, height, texels: Arc::new(buf.iter().flat_map(|c: &Color| vec![c.x, c.y, c.z]).collect()) }
        }
    };
    Some(SceneDesc { objects, lights, environment })
}

/// The scene on the device (one handle; `Renderer::sample` is called from one thread, the mutex only makes the
/// renderer `Sync` as the rayon path needs it to be)
pub struct GpuBackend {
    scene: Mutex<GpuScene>,
}

impl GpuBackend {
    /// Flatten and upload; `None` (-> CPU path) if the scene cannot be flattened or no device / library is usable
    pub fn new(scene: &Scene) -> Option<Self> {
        let desc = flatten_scene(scene)?;
        match GpuScene::new(&desc, 0) {
            Ok(s) => Some(Self { scene: Mutex::new(s) }),
            Err(e) => {
                eprintln!("rpt: GPU back-end unavailable ({}), rendering on the CPU", e);
                None
            }
        }
    }

    /// What `Renderer::sample` collects into `colors` (renderer.rs:118-127): W*H means, row-major, top row first
    pub fn render_batch(&self, r: &Renderer, iterations: u32, sample_base: u64) -> Vec<Color> {
        let params = RptRenderParams {
            width: r.width,
            height: r.height,
            max_bounces: r.max_bounces,
            iterations,
            exposure_value: r.exposure_value,
            seed: r.seed,
            sample_index_base: sample_base,
            tile_width: 32,
            tile_height: 8,
            part_index: 0,
            part_count: 1,
            precision_mode: 0, // RPT_PRECISION_F64_STRICT: IEEE f64, no FMA contraction — rpt's own arithmetic
            flags: 0,
            collective: 0, // RPT_COLLECTIVE_DEFAULT (only rptgpu_render_batch_reduce looks at it)
            _reserved0: 0,
        };
        let mut flat = vec![0.0f64; r.width as usize * r.height as usize * 3];
        self.scene
            .lock()
            .unwrap()
            .render_batch(&camera(&r.camera), &params, &mut flat)
            .unwrap_or_else(|e| panic!("rptgpu_render_batch: {}", e));
        flat.chunks_exact(3).map(|c| glm::vec3(c[0], c[1], c[2])).collect()
    }
}
