// host-side known-answer check of mapdn_amd/csrc/philox.hpp (compiled by tests/test_philox.py with g++): prints the Random123
// kat_vectors cases (philox4x32, 10 rounds) and, for argv = c0 c1 c2 c3 k0 k1 (hex), that block and its two u53 values.
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include "philox.hpp"

int main(int argc, char** argv) {
  uint32_t o[4];
  mapdn::philox4x32_10(0, 0, 0, 0, 0, 0, o);
  printf("philox0 %08x %08x %08x %08x\n", o[0], o[1], o[2], o[3]);
  mapdn::philox4x32_10(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, o);
  printf("philox1 %08x %08x %08x %08x\n", o[0], o[1], o[2], o[3]);
  mapdn::philox4x32_10(0x243f6a88u, 0x85a308d3u, 0x13198a2eu, 0x03707344u, 0xa4093822u, 0x299f31d0u, o);
  printf("philox2 %08x %08x %08x %08x\n", o[0], o[1], o[2], o[3]);
  if (argc == 7) {
    uint32_t a[6];
    for (int i = 0; i < 6; ++i) a[i] = (uint32_t)strtoul(argv[i + 1], nullptr, 16);
    mapdn::philox4x32_10(a[0], a[1], a[2], a[3], a[4], a[5], o);
    printf("block %08x %08x %08x %08x\n", o[0], o[1], o[2], o[3]);
    printf("u53 %.1f %.1f\n", mapdn::u53(o[0], o[1]), mapdn::u53(o[2], o[3]));
  }
  return 0;
}
