// Renders an imported asset through the C++ host mirror (include/rpt.hpp) on the GPU back-end:
//   obj_render <file.obj | file.stl | file.obj+file.mtl> [width height max_bounces num_samples seed out_prefix]
// The mesh (load_obj / load_stl, reference src/io.rs:27-74, 260-360) or the objects of an OBJ + MTL pair
// (load_obj_with_mtl, io.rs:83-148) stand on a diffuse floor under a sphere lamp and a point light, seen by the
// look_at camera the reference's asset examples use.  tests/test_gpu_io.py builds the same scene with the Python
// mirror's importers and the oracle, and compares the frames bit for bit.
#include <cstdlib>
#include <string>
#include "rpt.hpp"
#include "dump.hpp"
using namespace rpt;

int main(int argc, char** argv) {
  if (argc < 2) { std::fprintf(stderr, "usage: obj_render asset [w h bounces spp seed prefix]\n"); return 2; }
  std::string asset = argv[1];
  Scene scene;
  try {
    size_t plus = asset.find('+');
    if (plus != std::string::npos) {
      for (const Object& o : load_obj_with_mtl(asset.substr(0, plus), asset.substr(plus + 1))) scene.add(o);
    } else if (asset.size() > 4 && asset.substr(asset.size() - 4) == ".stl") {
      scene.add(Object(load_stl(asset)).material(Material::specular(hex_color(0xB7CA79), 0.2)));
    } else {
      scene.add(Object(load_obj(asset)).material(Material::specular(hex_color(0xB7CA79), 0.2)));
    }
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  scene.add(Object(plane({0.0, 1.0, 0.0}, -1.0)).material(Material::diffuse(hex_color(0xAAAAAA))));
  scene.add(Light::Ambient({0.02, 0.02, 0.02}));
  scene.add(Light::Object(Object(sphere().scale({1.5, 1.5, 1.5}).translate({0.0, 8.0, 3.0})).material(Material::light({1.0, 1.0, 1.0}, 60.0))));
  scene.add(Light::Point({20.0, 20.0, 20.0}, {-3.0, 4.0, 4.0}));
  Camera camera = Camera::look_at({-2.5, 3.0, 5.5}, {0.0, 0.0, 0.0}, {0.0, 1.0, 0.0}, 3.14159265358979323846 / 5.0);
  return run_example(scene, camera, argc - 1, argv + 1, 128, 72, 4, 4);
}
