#!/usr/bin/env python3
"""Generates rust/rpt.patch: the changes a maintainer applies to ekzhang/rpt so that

  * `cargo build --features gpu` renders through librptgpu.so (Renderer::sample -> rptgpu_render_batch, every shape of
    the closed device set gets a `flatten`), with rpt itself still `#![forbid(unsafe_code)]`;
  * `cargo run --release --features philox --example dump_golden` renders with the SAME Philox4x32-10 stream the
    back-end and its CPU oracle use (instead of entropy-seeded StdRng, src/renderer.rs:121) and dumps f64 frames that
    `scripts/compare_rust_golden.py` compares with the oracle — the step that turns "parity unpinned" into "pinned"
    for whoever has a Rust toolchain.

Usage (needs the reference checkout; only the PATCH is committed, never reference sources):
    python rust/make_patch.py /root/reference [output file, default rust/rpt.patch]
The new files of the patch (src/rng.rs, src/gpu.rs, examples/dump_golden.rs) live next to this script under
rust/rpt_additions/ and are original code of this repository.
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ADD = os.path.join(HERE, "rpt_additions")


def sub1(text, old, new, path):
    assert old in text, "%s: anchor not found: %r" % (path, old[:60])
    return text.replace(old, new, 1)


def edit(root, rel, fn):
    p = os.path.join(root, rel)
    s = open(p).read()
    t = fn(s, rel)
    assert t != s, rel
    open(p, "w").write(t)


def rng_type(s, rel):
    """`&mut StdRng` -> `&mut PathRng` in every signature; imports adjusted."""
    s = s.replace("use rand::{rngs::StdRng, Rng, SeedableRng};", "use rand::Rng;")
    s = s.replace("use rand::{distributions::Uniform, rngs::StdRng, Rng};", "use rand::{distributions::Uniform, Rng};")
    s = s.replace("use rand::{rngs::StdRng, Rng};", "use rand::Rng;")
    s = s.replace("use rand::rngs::StdRng;\n", "")
    s = re.sub(r"\bStdRng\b", "PathRng", s)
    # first `use crate::` / `use super::` line gets the new import in front of it
    m = re.search(r"^use (crate|super)::", s, flags=re.M)
    ins = "use crate::rng::PathRng;\n"
    return s[:m.start()] + ins + s[m.start():] if m else ins + s


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    dest = sys.argv[2] if len(sys.argv) > 2 else os.path.join(HERE, "rpt.patch")  # (tests regenerate into a temp file)
    tmp = tempfile.mkdtemp(prefix="rpt_patch_")
    a, b = os.path.join(tmp, "a"), os.path.join(tmp, "b")
    for d in (a, b):
        os.makedirs(d)
        shutil.copy(os.path.join(ref, "Cargo.toml"), d)
        shutil.copytree(os.path.join(ref, "src"), os.path.join(d, "src"))
        os.makedirs(os.path.join(d, "examples"))

    # ---- Cargo.toml
    edit(b, "Cargo.toml", lambda s, r: sub1(s, 'rayon = "1.5.0"\n', '''rayon = "1.5.0"
rpt-gpu-sys = { path = "../rpt-gpu-sys", optional = true }

[features]
# render through librptgpu.so (the MI355X back-end) instead of the rayon loop of Renderer::sample
gpu = ["rpt-gpu-sys"]
# Philox4x32-10 keyed by (seed; pixel, sample) instead of one entropy-seeded StdRng per row: the stream of the
# back-end and of its CPU oracle, so that equal seeds give comparable images
philox = []
''', r))

    # ---- the RNG type
    for rel in ("src/kdtree.rs", "src/shape/mesh.rs", "src/shape/monomial_surface.rs", "src/shape/sphere.rs",
                "src/shape/plane.rs", "src/shape/cube.rs", "src/camera.rs", "src/material.rs", "src/shape.rs",
                "src/light.rs", "src/renderer.rs"):
        edit(b, rel, rng_type)

    # ---- lib.rs
    def lib(s, r):
        s = sub1(s, "pub use renderer::*;\n", "pub use renderer::*;\npub use rng::*;\n", r)
        s = sub1(s, "mod renderer;\n", "mod renderer;\nmod rng;\n", r)
        s = sub1(s, "pub use environment::*;\n", "pub use environment::*;\n#[cfg(feature = \"gpu\")]\npub use gpu::*;\n", r)
        s = sub1(s, "mod environment;\n", "mod environment;\n#[cfg(feature = \"gpu\")]\nmod gpu;\n", r)
        return s
    edit(b, "src/lib.rs", lib)

    # ---- shape.rs: the flatten hook on the trait, forwarding impls, Transformed
    FL = "#[cfg(feature = \"gpu\")]\n"
    def shape(s, r):
        s = sub1(s, "    fn sample(&self, target: &glm::DVec3, rng: &mut PathRng) -> (glm::DVec3, glm::DVec3, f64);\n}\n",
                 "    fn sample(&self, target: &glm::DVec3, rng: &mut PathRng) -> (glm::DVec3, glm::DVec3, f64);\n\n"
                 "    /// Describe this shape to the GPU back-end; `None` = outside the closed device set, the renderer\n"
                 "    /// then keeps the CPU path for the whole scene\n"
                 "    #[cfg(feature = \"gpu\")]\n"
                 "    fn flatten(&self, _arena: &mut crate::gpu::FlatArena) -> Option<rpt_gpu_sys::ShapeDesc> {\n"
                 "        None\n    }\n}\n", r)
        fwd = ("\n    #[cfg(feature = \"gpu\")]\n"
               "    fn flatten(&self, arena: &mut crate::gpu::FlatArena) -> Option<rpt_gpu_sys::ShapeDesc> {\n"
               "        self.as_ref().flatten(arena)\n    }\n}\n")
        # Box<T> and Arc<T> forwarding impls (shape.rs:27-45)
        for _ in range(2):
            s = sub1(s, "        self.as_ref().sample(target, rng)\n    }\n}\n", "        self.as_ref().sample(target, rng)\n    }@@FWD@@", r)
        s = s.replace("@@FWD@@", fwd)
        s = sub1(s, "            p / parallelepiped_base, // divide PDF by the area scale factor\n        )\n    }\n}\n",
                 "            p / parallelepiped_base, // divide PDF by the area scale factor\n        )\n    }\n\n"
                 "    /// The five fields as they are (nalgebra stores column-major, the ABI's convention)\n"
                 "    #[cfg(feature = \"gpu\")]\n"
                 "    fn flatten(&self, arena: &mut crate::gpu::FlatArena) -> Option<rpt_gpu_sys::ShapeDesc> {\n"
                 "        let mut xf = rpt_gpu_sys::RptTransform {\n"
                 "            transform: [0.0; 16],\n            linear: [0.0; 9],\n            inverse_transform: [0.0; 16],\n"
                 "            normal_transform: [0.0; 9],\n            scale: self.scale,\n        };\n"
                 "        xf.transform.copy_from_slice(self.transform.as_slice());\n"
                 "        xf.linear.copy_from_slice(self.linear.as_slice());\n"
                 "        xf.inverse_transform.copy_from_slice(self.inverse_transform.as_slice());\n"
                 "        xf.normal_transform.copy_from_slice(self.normal_transform.as_slice());\n"
                 "        Some(rpt_gpu_sys::ShapeDesc::Transformed {\n"
                 "            inner: Box::new(self.shape.flatten(arena)?),\n            xf,\n        })\n    }\n}\n", r)
        return s
    edit(b, "src/shape.rs", shape)

    # ---- the unit primitives
    def after_sample_fn(s, r, body):
        """append a flatten method at the end of the `impl Shape for X` block: anchor = the closing of `fn sample`"""
        m = re.search(r"impl Shape for \w+ \{", s)
        assert m, r
        # the impl block ends at the first "\n}\n" after its `fn sample`
        k = s.index("fn sample", m.end())
        e = s.index("\n}\n", k)
        return s[:e] + "\n\n    #[cfg(feature = \"gpu\")]\n    fn flatten(&self, _arena: &mut crate::gpu::FlatArena) -> Option<rpt_gpu_sys::ShapeDesc> {\n        " + body + "\n    }" + s[e:]
    edit(b, "src/shape/sphere.rs", lambda s, r: after_sample_fn(rng_keep(s), r, "Some(rpt_gpu_sys::ShapeDesc::Sphere)"))
    edit(b, "src/shape/cube.rs", lambda s, r: after_sample_fn(rng_keep(s), r, "Some(rpt_gpu_sys::ShapeDesc::Cube)"))
    edit(b, "src/shape/plane.rs", lambda s, r: after_sample_fn(rng_keep(s), r,
         "Some(rpt_gpu_sys::ShapeDesc::Plane {\n            normal: [self.normal.x, self.normal.y, self.normal.z],\n            value: self.value,\n        })"))
    edit(b, "src/shape/monomial_surface.rs", lambda s, r: after_sample_fn(rng_keep(s), r,
         "Some(rpt_gpu_sys::ShapeDesc::Monomial {\n            height: self.height,\n            exp: self.exp,\n        })"))

    # ---- Triangle / Mesh / KdTree
    def mesh(s, r):
        s = sub1(s, "impl Bounded for Triangle {\n    fn bounding_box(&self) -> BoundingBox {\n        BoundingBox {\n"
                    "            p_min: glm::min3(&self.v1, &self.v2, &self.v3),\n"
                    "            p_max: glm::max3(&self.v1, &self.v2, &self.v3),\n        }\n    }\n}\n",
                 "impl Bounded for Triangle {\n    fn bounding_box(&self) -> BoundingBox {\n        BoundingBox {\n"
                 "            p_min: glm::min3(&self.v1, &self.v2, &self.v3),\n"
                 "            p_max: glm::max3(&self.v1, &self.v2, &self.v3),\n        }\n    }\n\n"
                 "    /// A kd-tree of triangles is a Mesh: its triangles go to the device as one array, shared by every\n"
                 "    /// instance of the same `Arc<Mesh>` (the arena keys them by address)\n"
                 "    #[cfg(feature = \"gpu\")]\n"
                 "    fn flatten_collection(objects: &[Self], arena: &mut crate::gpu::FlatArena) -> Option<rpt_gpu_sys::ShapeDesc> {\n"
                 "        Some(rpt_gpu_sys::ShapeDesc::Mesh(arena.mesh(objects)))\n    }\n}\n", r)
        return s
    edit(b, "src/shape/mesh.rs", mesh)

    def kdtree(s, r):
        s = sub1(s, "    /// Returns the shape's bounding box\n    fn bounding_box(&self) -> BoundingBox;\n}\n",
                 "    /// Returns the shape's bounding box\n    fn bounding_box(&self) -> BoundingBox;\n\n"
                 "    /// How a `KdTree<Self>` describes itself to the GPU back-end: by default a GROUP of individually\n"
                 "    /// flattened children (`KdTree<Box<dyn Bounded>>`); `Triangle` overrides it with a MESH\n"
                 "    #[cfg(feature = \"gpu\")]\n"
                 "    fn flatten_collection(objects: &[Self], arena: &mut crate::gpu::FlatArena) -> Option<rpt_gpu_sys::ShapeDesc>\n"
                 "    where\n        Self: Sized,\n    {\n"
                 "        objects\n            .iter()\n            .map(|o| o.flatten(arena))\n            .collect::<Option<Vec<_>>>()\n"
                 "            .map(rpt_gpu_sys::ShapeDesc::Group)\n    }\n}\n", r)
        s = sub1(s, "        (v, n, p / (num as f64))\n    }\n}\n",
                 "        (v, n, p / (num as f64))\n    }\n\n"
                 "    #[cfg(feature = \"gpu\")]\n"
                 "    fn flatten(&self, arena: &mut crate::gpu::FlatArena) -> Option<rpt_gpu_sys::ShapeDesc> {\n"
                 "        T::flatten_collection(&self.objects, arena)\n    }\n}\n", r)
        return s
    edit(b, "src/kdtree.rs", kdtree)

    # ---- renderer.rs: seed, batch counter, the GPU branch of sample(), the Philox branch of get_color()
    def renderer(s, r):
        s = sub1(s, "    /// Number of random paths traced per pixel\n    pub num_samples: u32,\n}\n",
                 "    /// Number of random paths traced per pixel\n    pub num_samples: u32,\n\n"
                 "    /// Seed of the path streams (used by the `philox` and `gpu` features; the default build is\n"
                 "    /// entropy-seeded as before)\n    pub seed: u64,\n\n"
                 "    /// Samples per pixel handed out so far: batch k of `iterative_render` continues the streams of batch k-1\n"
                 "    samples_done: std::sync::atomic::AtomicU64,\n\n"
                 "    /// The scene on the device, if every shape is in the back-end's closed set\n"
                 "    #[cfg(feature = \"gpu\")]\n    gpu: Option<crate::gpu::GpuBackend>,\n}\n", r)
        s = sub1(s, "            max_bounces: 0,\n            num_samples: 1,\n        }\n",
                 "            max_bounces: 0,\n            num_samples: 1,\n            seed: 0x5250_5447,\n"
                 "            samples_done: std::sync::atomic::AtomicU64::new(0),\n"
                 "            #[cfg(feature = \"gpu\")]\n            gpu: crate::gpu::GpuBackend::new(scene),\n        }\n", r)
        s = sub1(s, "    /// Render the scene by path tracing\n",
                 "    /// Set the seed of the path streams\n    pub fn seed(mut self, seed: u64) -> Self {\n        self.seed = seed;\n        self\n    }\n\n"
                 "    /// Render the scene by path tracing\n", r)
        s = sub1(s, "    fn sample(&self, iterations: u32, buffer: &mut Buffer) {\n",
                 "    fn sample(&self, iterations: u32, buffer: &mut Buffer) {\n"
                 "        let sample_base = self\n            .samples_done\n            .fetch_add(u64::from(iterations), std::sync::atomic::Ordering::Relaxed);\n"
                 "        #[cfg(feature = \"gpu\")]\n        if let Some(gpu) = &self.gpu {\n"
                 "            // the whole body of this function on the device: rptgpu_render_batch (include/rpt_gpu.h)\n"
                 "            let colors = gpu.render_batch(self, iterations, sample_base);\n"
                 "            buffer.add_samples(&colors);\n            return;\n        }\n", r)
        s = sub1(s, "                let mut rng = PathRng::from_entropy();\n",
                 "                #[cfg(not(feature = \"philox\"))]\n                let mut rng = <PathRng as rand::SeedableRng>::from_entropy();\n"
                 "                #[cfg(feature = \"philox\")]\n                let mut rng = PathRng::for_sample(self.seed, 0, sample_base);\n", r)
        s = sub1(s, "        for _ in 0..iterations {\n            let dx = rng.gen_range((-1.0 / dim)..(1.0 / dim));\n",
                 "        for _i in 0..iterations {\n"
                 "            // one stream per (pixel, sample): the image no longer depends on how rows are scheduled\n"
                 "            #[cfg(feature = \"philox\")]\n"
                 "            rng.restart(y * self.width + x, rng.sample_base() + u64::from(_i));\n"
                 "            let dx = rng.gen_range((-1.0 / dim)..(1.0 / dim));\n", r)
        return s
    edit(b, "src/renderer.rs", renderer)

    # ---- accessors the new code needs
    edit(b, "src/environment.rs", lambda s, r: sub1(s, "        Self { width, height, buf }\n    }\n",
         "        Self { width, height, buf }\n    }\n\n"
         "    /// Width, height and the row-major pixel buffer (for back-ends that upload the image)\n"
         "    pub fn raw(&self) -> (u32, u32, &[Color]) {\n        (self.width, self.height, &self.buf)\n    }\n", r))
    edit(b, "src/buffer.rs", lambda s, r: sub1(s, "    /// Converts the current buffer to an image\n",
         "    /// The last batch of per-pixel means as flat f64 triples, row-major, top row first (what\n"
         "    /// `Renderer::sample` produced; for golden-vector dumps)\n"
         "    pub fn raw_means(&self) -> Vec<f64> {\n"
         "        self.samples\n            .iter()\n            .flat_map(|s| {\n"
         "                let c = s.last().copied().unwrap_or_else(|| glm::vec3(0.0, 0.0, 0.0));\n"
         "                vec![c.x, c.y, c.z]\n            })\n            .collect()\n    }\n\n"
         "    /// Converts the current buffer to an image\n", r))

    # ---- new files
    shutil.copy(os.path.join(ADD, "rng.rs"), os.path.join(b, "src", "rng.rs"))
    shutil.copy(os.path.join(ADD, "gpu.rs"), os.path.join(b, "src", "gpu.rs"))
    shutil.copy(os.path.join(ADD, "dump_golden.rs"), os.path.join(b, "examples", "dump_golden.rs"))

    out = subprocess.run(["diff", "-ruN", "a", "b"], cwd=tmp, capture_output=True, text=True).stdout
    out = re.sub(r"^(---|\+\+\+) (\S+)\t.*$", r"\1 \2", out, flags=re.M)  # no timestamps
    open(dest, "w").write(out)
    print("wrote %s: %d lines, %d files" % (dest, out.count("\n"), out.count("\ndiff -ruN") + 1))
    shutil.rmtree(tmp)


def rng_keep(s):
    return s


if __name__ == "__main__":
    main()
