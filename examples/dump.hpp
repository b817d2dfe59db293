// shared driver of the example programs: render through rpt::Renderer, report the time the way
// the reference's examples do (examples/cornell.rs:82-96), save a PPM and, for the parity
// tests, the raw f64 batch means.
#pragma once
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include "rpt.hpp"

inline int run_example(const rpt::Scene& scene, const rpt::Camera& camera, int argc, char** argv, uint32_t w,
                       uint32_t h, uint32_t bounces, uint32_t spp) {
  uint64_t seed = 0x52505447;
  std::string prefix = "output";
  if (argc > 1) w = (uint32_t)std::atoi(argv[1]);
  if (argc > 2) h = (uint32_t)std::atoi(argv[2]);
  if (argc > 3) bounces = (uint32_t)std::atoi(argv[3]);
  if (argc > 4) spp = (uint32_t)std::atoi(argv[4]);
  if (argc > 5) seed = std::strtoull(argv[5], nullptr, 0);
  if (argc > 6) prefix = argv[6];
  try {
    rpt::Renderer renderer(scene, camera);
    renderer.width(w).height(h).max_bounces(bounces).num_samples(spp).seed(seed);
    rpt::Buffer buffer(w, h);
    auto t0 = std::chrono::steady_clock::now();
    renderer.sample(spp, buffer); // what render() does (renderer.rs:96-100), keeping the Buffer
    double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    std::printf("Finished %u samples per pixel at %ux%u, took %.1f ms (%.2f Msamples/s incl. scene upload)\n", spp, w, h,
                ms, (double)w * h * spp / ms / 1e3);
    buffer.image().save_ppm(prefix + ".ppm");
    FILE* f = std::fopen((prefix + ".f64").c_str(), "wb");
    std::fwrite(buffer.samples[0].data(), sizeof(double), buffer.samples[0].size(), f);
    std::fclose(f);
  } catch (const rpt::GpuError& e) {
    std::fprintf(stderr, "rptgpu error %d: %s\n", e.code, e.what());
    return 2;
  }
  return 0;
}
