//! Golden vectors of the REFERENCE for the MI355X back-end's oracle.
//!
//!     cargo run --release --features philox --example dump_golden -- out_dir
//!
//! Renders small versions of examples/sphere.rs and examples/cornell.rs — and, when `examples/teapot.obj` is where the
//! crate keeps it, a scene around that mesh (kd-tree build + traversal, metal, glass, delta lights) — on the CPU with the Philox stream
//! (seed, pixel, sample) and writes the W*H*3 f64 means (little endian, row-major, top row first — the `colors`
//! vector of Renderer::sample, renderer.rs:118-127) to `<out_dir>/<name>.f64` plus a `<name>.txt` with the
//! parameters.  `python scripts/compare_rust_golden.py out_dir` in the back-end repository renders the same
//! configurations with the oracle and reports per-channel differences (expected: 0 up to the platform libm's
//! last-ulp differences in exp/ln/atan/sin_cos/acos/atan2).
use std::fs;
use std::io::Write;

use rpt::*;

fn dump(dir: &str, name: &str, scene: &Scene, camera: Camera, w: u32, h: u32, bounces: u32, spp: u32, seed: u64) {
    let mut frame: Vec<f64> = Vec::new();
    Renderer::new(scene, camera)
        .width(w)
        .height(h)
        .max_bounces(bounces)
        .num_samples(spp)
        .seed(seed)
        .iterative_render(spp, |_, buffer| {
            // one batch: the buffer holds exactly the means of Renderer::sample
            frame = buffer.raw_means();
        });
    let mut f = fs::File::create(format!("{}/{}.f64", dir, name)).unwrap();
    for v in &frame {
        f.write_all(&v.to_le_bytes()).unwrap();
    }
    fs::write(
        format!("{}/{}.txt", dir, name),
        format!("width {}\nheight {}\nmax_bounces {}\niterations {}\nseed {}\n", w, h, bounces, spp, seed),
    )
    .unwrap();
    println!("{}: {}x{} B={} spp={} seed={}", name, w, h, bounces, spp, seed);
}

fn main() {
    let dir = std::env::args().nth(1).unwrap_or_else(|| "golden_out".into());
    fs::create_dir_all(&dir).unwrap();

    // examples/sphere.rs
    let mut scene = Scene::new();
    scene.add(Object::new(sphere()));
    scene.add(Object::new(plane(glm::vec3(0.0, 1.0, 0.0), -1.0)).material(Material::diffuse(hex_color(0xAAAAAA))));
    scene.add(Light::Object(
        Object::new(sphere().scale(&glm::vec3(2.0, 2.0, 2.0)).translate(&glm::vec3(0.0, 12.0, 0.0)))
            .material(Material::light(hex_color(0xFFFFFF), 40.0)),
    ));
    let camera = Camera::look_at(glm::vec3(-2.5, 4.0, 6.5), glm::vec3(0.0, -0.25, 0.0), glm::vec3(0.0, 1.0, 0.0), std::f64::consts::FRAC_PI_4);
    dump(&dir, "sphere", &scene, camera, 64, 36, 2, 8, 101);

    // examples/cornell.rs
    let mut scene = Scene::new();
    let white = Material::diffuse(hex_color(0xAAAAAA));
    let red = Material::diffuse(hex_color(0xBC0000));
    let green = Material::diffuse(hex_color(0x00BC00));
    let quad = |v: [[f64; 3]; 4]| polygon(&v.iter().map(|p| glm::vec3(p[0], p[1], p[2])).collect::<Vec<_>>());
    scene.add(Object::new(quad([[0.0, 0.0, 0.0], [0.0, 0.0, 559.2], [556.0, 0.0, 559.2], [556.0, 0.0, 0.0]])).material(white));
    scene.add(Object::new(quad([[0.0, 548.9, 0.0], [556.0, 548.9, 0.0], [556.0, 548.9, 559.2], [0.0, 548.9, 559.2]])).material(white));
    scene.add(Object::new(quad([[0.0, 0.0, 559.2], [0.0, 548.9, 559.2], [556.0, 548.9, 559.2], [556.0, 0.0, 559.2]])).material(white));
    scene.add(Object::new(quad([[556.0, 0.0, 0.0], [556.0, 0.0, 559.2], [556.0, 548.9, 559.2], [556.0, 548.9, 0.0]])).material(red));
    scene.add(Object::new(quad([[0.0, 0.0, 0.0], [0.0, 548.9, 0.0], [0.0, 548.9, 559.2], [0.0, 0.0, 559.2]])).material(green));
    let tau = 2.0 * std::f64::consts::PI;
    scene.add(
        Object::new(cube().scale(&glm::vec3(165.0, 330.0, 165.0)).rotate_y(tau * (-253.0 / 360.0)).translate(&glm::vec3(368.0, 165.0, 351.0)))
            .material(white),
    );
    scene.add(
        Object::new(cube().scale(&glm::vec3(165.0, 165.0, 165.0)).rotate_y(tau * (-197.0 / 360.0)).translate(&glm::vec3(185.0, 82.5, 169.0)))
            .material(white),
    );
    scene.add(Light::Object(
        Object::new(quad([[343.0, 548.8, 227.0], [343.0, 548.8, 332.0], [213.0, 548.8, 332.0], [213.0, 548.8, 227.0]]))
            .material(Material::light(hex_color(0xFFFEFA), 100.0)),
    ));
    let camera = Camera { eye: glm::vec3(278.0, 273.0, -800.0), direction: glm::vec3(0.0, 0.0, 1.0), up: glm::vec3(0.0, 1.0, 0.0), fov: 0.686, aperture: 0.0, focal_distance: 0.0 };
    dump(&dir, "cornell", &scene, camera, 64, 36, 8, 8, 102);

    // examples/teapot.rs's mesh and placement (KdTree::new + intersect_subtree on the crate's own 2 256-face asset), with a
    // glass sphere beside it (the transparent branches of bsdf / sample_f), six bounces; run from the crate's root
    if let Ok(file) = fs::File::open("examples/teapot.obj") {
        let mut scene = Scene::new();
        scene.add(
            Object::new(load_obj(file).unwrap().scale(&glm::vec3(0.5, 0.5, 0.5)).translate(&glm::vec3(0.0, -1.0, 0.0)))
                .material(Material::metallic(hex_color(0xff0000), 0.4)),
        );
        scene.add(
            Object::new(sphere().scale(&glm::vec3(0.4, 0.4, 0.4)).translate(&glm::vec3(1.3, -0.6, 0.8)))
                .material(Material::clear(1.5, 0.02)),
        );
        scene.add(Object::new(plane(glm::vec3(0.0, 1.0, 0.0), -1.0)).material(Material::diffuse(hex_color(0xaaaaaa))));
        scene.add(Light::Ambient(glm::vec3(0.02, 0.02, 0.02)));
        scene.add(Light::Point(glm::vec3(60.0, 60.0, 60.0), glm::vec3(0.0, 5.0, 5.0)));
        scene.environment = Environment::Color(glm::vec3(0.3, 0.4, 0.6));
        dump(&dir, "teapot", &scene, Camera::default(), 64, 64, 6, 8, 103);
    } else {
        println!("teapot: examples/teapot.obj not found (run from the crate's root): skipped");
    }
}
