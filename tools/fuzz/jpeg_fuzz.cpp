#include <cstdio>
#include <fstream>
#include <iterator>
#include <random>
#include "../../raytracer_amd/csrc/host/jpeg_baseline.hpp"  // g++ -std=c++17 -O1 -g -fsanitize=address,undefined tools/fuzz/jpeg_fuzz.cpp && ./a.out tests/golden/earthmap.jpg
int main(int argc, char** argv) {
  std::ifstream f(argv[1], std::ios::binary);
  std::vector<uint8_t> data((std::istreambuf_iterator<char>(f)), {});
  std::mt19937 rng(1);
  int ok = 0, refused = 0;
  for (int it = 0; it < 4000; ++it) {
    std::vector<uint8_t> b = data;
    int mode = it % 4;
    if (mode == 0) b.resize(rng() % b.size());
    else if (mode == 1) for (int k = 0; k < 8; ++k) b[rng() % b.size()] = (uint8_t)rng();
    else if (mode == 2) for (int k = 0; k < 3; ++k) b[rng() % 700] = (uint8_t)rng();   // headers
    else { size_t at = 700 + rng() % (b.size() - 800); for (int k = 0; k < 64; ++k) b[at + k] = (uint8_t)rng(); }
    try {
      uint32_t w, h;
      rth::jpeg::Decoder(b.data(), b.size()).decode(w, h);
      ++ok;
    } catch (const std::exception&) { ++refused; }
  }
  printf("decoded %d refused %d\n", ok, refused);
}
