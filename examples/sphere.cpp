// examples/sphere.rs of the reference (lines 3-35), transcribed against include/rpt.hpp.
// usage: sphere [width height max_bounces num_samples seed out_prefix]
#include <cstdlib>
#include "rpt.hpp"
#include "dump.hpp"
using namespace rpt;

int main(int argc, char** argv) {
  Scene scene;
  scene.add(Object(sphere())); // default red material
  scene.add(Object(plane({0.0, 1.0, 0.0}, -1.0)).material(Material::diffuse(hex_color(0xAAAAAA))));
  scene.add(Light::Object(Object(sphere().scale({2.0, 2.0, 2.0}).translate({0.0, 12.0, 0.0}))
                              .material(Material::light(hex_color(0xFFFFFF), 40.0))));
  Camera camera = Camera::look_at({-2.5, 4.0, 6.5}, {0.0, -0.25, 0.0}, {0.0, 1.0, 0.0}, 3.14159265358979323846 / 4.0);
  return run_example(scene, camera, argc, argv, 960, 540, 2, 100);
}
