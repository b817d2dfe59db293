// Loads an OBJ or STL file with the C++ mirror's loaders (include/rpt.hpp, reference src/io.rs) and prints the
// triangle count and every coordinate as a hex float, for comparison with the Python mirror (tests/test_io.py).
#include <cstdio>
#include <string>

#include "rpt.hpp"

int main(int argc, char** argv) {
  if (argc < 2) { std::fprintf(stderr, "usage: io_check file.{obj,stl}\n"); return 2; }
  std::string path = argv[1];
  try {
    rpt::Shape m = path.size() > 4 && path.substr(path.size() - 4) == ".stl" ? rpt::load_stl(path) : rpt::load_obj(path);
    const auto& t = m.node->triangles;
    std::printf("%zu\n", t.size());
    for (const RptTriangle& r : t) {
      const double* p = r.v1;
      for (int k = 0; k < 18; k++) std::printf("%a%c", p[k], k == 17 ? '\n' : ' ');
    }
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
