// examples/cornell.rs of the reference (lines 9-102), transcribed against include/rpt.hpp.
// usage: cornell [width height max_bounces num_samples seed out_prefix]
#include <cstdlib>
#include "rpt.hpp"
#include "dump.hpp"
using namespace rpt;

int main(int argc, char** argv) {
  Scene scene;
  Camera camera;
  camera.eye = {278.0, 273.0, -800.0};
  camera.direction = {0.0, 0.0, 1.0};
  camera.up = {0.0, 1.0, 0.0};
  camera.fov = 0.686;

  Material white = Material::diffuse(hex_color(0xAAAAAA));
  Material red = Material::diffuse(hex_color(0xBC0000));
  Material green = Material::diffuse(hex_color(0x00BC00));
  Material light_mtl = Material::light(hex_color(0xFFFEFA), 100.0); // 6500 K

  Shape floor = polygon({{0.0, 0.0, 0.0}, {0.0, 0.0, 559.2}, {556.0, 0.0, 559.2}, {556.0, 0.0, 0.0}});
  Shape ceiling = polygon({{0.0, 548.9, 0.0}, {556.0, 548.9, 0.0}, {556.0, 548.9, 559.2}, {0.0, 548.9, 559.2}});
  Shape light_rect = polygon({{343.0, 548.8, 227.0}, {343.0, 548.8, 332.0}, {213.0, 548.8, 332.0}, {213.0, 548.8, 227.0}});
  Shape back_wall = polygon({{0.0, 0.0, 559.2}, {0.0, 548.9, 559.2}, {556.0, 548.9, 559.2}, {556.0, 0.0, 559.2}});
  Shape right_wall = polygon({{0.0, 0.0, 0.0}, {0.0, 548.9, 0.0}, {0.0, 548.9, 559.2}, {0.0, 0.0, 559.2}});
  Shape left_wall = polygon({{556.0, 0.0, 0.0}, {556.0, 0.0, 559.2}, {556.0, 548.9, 559.2}, {556.0, 548.9, 0.0}});

  const double two_pi = 2.0 * 3.14159265358979323846;
  Shape large_box = cube().scale({165.0, 330.0, 165.0}).rotate_y(two_pi * (-253.0 / 360.0)).translate({368.0, 165.0, 351.0});
  Shape small_box = cube().scale({165.0, 165.0, 165.0}).rotate_y(two_pi * (-197.0 / 360.0)).translate({185.0, 82.5, 169.0});

  scene.add(Object(floor).material(white));
  scene.add(Object(ceiling).material(white));
  scene.add(Object(back_wall).material(white));
  scene.add(Object(left_wall).material(red));
  scene.add(Object(right_wall).material(green));
  scene.add(Object(large_box).material(white));
  scene.add(Object(small_box).material(white));
  scene.add(Light::Object(Object(light_rect).material(light_mtl)));
  return run_example(scene, camera, argc, argv, 1024, 1024, 2, 100);
}
