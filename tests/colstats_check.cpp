// host-side check of mapdn_amd/csrc/colstats.hpp (compiled by tests/test_colstats.py with g++): reads a row-major table
// (int64 T, int64 ncol, T * ncol doubles) and prints every column's std / 100 as a hex-float, to be compared BIT FOR BIT with
// numpy's `np.asfortranarray(table).std(axis=0) / 100` — the reference's `DataFrame.values.std(axis=0) / 100.0`
// (voltage_control_env.py:70-72).
#include <cstdint>
#include <cstdio>
#include <vector>
#include "colstats.hpp"

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 3;
  int64_t T = 0, ncol = 0;
  if (fread(&T, 8, 1, f) != 1 || fread(&ncol, 8, 1, f) != 1) return 4;
  std::vector<double> tab((size_t)T * (size_t)ncol);
  if (fread(tab.data(), 8, tab.size(), f) != tab.size()) return 5;
  fclose(f);
  std::vector<double> sd;
  mapdn::column_std(tab.data(), T, (int)ncol, 100.0, sd);
  for (double v : sd) printf("%a\n", v);
  return 0;
}
