// nr_inst_list.hpp — the (W, L, HL, GL, RES) instantiations of k_nr_tree (nr_tree.hpp) that are compiled in, as four PARTS
// that the build compiles in parallel (nr_inst.hip with -DNR_INST_PART=0..3).
//   W waves per workgroup, L envs per workgroup;  HL / GL: the h / G factors are LDS-resident;
//   RES: residency of the step records / flat-start constants as a compile-time fact — 1 both in LDS, 2 neither, 3 records
//        only, 0 decided per handle at run time (the generic body: any residency the host settles on).
// Specialised (RES != 0) entries exist for what choose_nr_geometry (capi.hip) returns on the three MAPDN feeders; every
// (W, L) pair the chooser or a caller's mapdn_env_config.nr_waves / nr_lanes may ask for has the generic bodies as well.
// A geometry that is not in the list is refused at mapdn_create ("not compiled in"), never silently replaced.
#pragma once

// part 0: the 141-bus class (48 <= n < 200), fat layouts: 16 workers on four waves
#define NR_INSTS_0(X) \
  X(4, 16, true, false, 1) X(4, 16, true, false, 3) X(4, 16, true, false, 0) X(4, 16, true, true, 0) X(4, 16, false, false, 0)
// part 1: small feeders (n < 48: one wave) and the lean layout of the 141-bus class
#define NR_INSTS_1(X) \
  X(1, 16, true, true, 1) X(1, 16, true, true, 0) X(1, 16, true, false, 0) X(1, 16, false, false, 0) \
  X(2, 16, false, false, 2) X(2, 16, false, false, 0) X(2, 16, true, false, 0) X(2, 16, true, true, 0)
// part 2: the 322-bus class: 8 envs per workgroup (32 workers), 16 beyond one round of workgroups
#define NR_INSTS_2(X) \
  X(4, 8, true, false, 3) X(4, 8, true, false, 2) X(4, 8, true, false, 0) X(4, 8, true, true, 0) X(4, 8, false, false, 0) \
  X(4, 16, false, false, 2)
// part 3: further pairs a caller may force (tests: every geometry gives the same bits)
#define NR_INSTS_3(X) \
  X(1, 8, true, true, 0) X(1, 8, true, false, 0) X(1, 8, false, false, 0) \
  X(2, 8, true, true, 0) X(2, 8, true, false, 0) X(2, 8, false, false, 0) \
  X(1, 32, false, false, 0) X(1, 32, true, false, 0) X(1, 32, true, true, 0) \
  X(4, 4, true, false, 0) X(4, 4, true, true, 0) X(4, 4, false, false, 0) \
  X(2, 4, true, false, 0) X(2, 4, true, true, 0) X(2, 4, false, false, 0) \
  X(1, 4, true, false, 0) X(1, 4, true, true, 0)

namespace mapdn {
struct NrInst { int W, L, HL, GL, RES; const void* fn; };
enum { NR_INST_PARTS = 4 };
extern const NrInst nr_insts_0[]; extern const int nr_n_insts_0;
extern const NrInst nr_insts_1[]; extern const int nr_n_insts_1;
extern const NrInst nr_insts_2[]; extern const int nr_n_insts_2;
extern const NrInst nr_insts_3[]; extern const int nr_n_insts_3;
}  // namespace mapdn
