// nr_inst.hip — one PART of the k_nr_tree instantiations (nr_inst_list.hpp); compiled once per part, in parallel:
//   hipcc -c -DNR_INST_PART=<p> nr_inst.hip -o nr_inst_<p>.o
#include "nr_tree.hpp"
#include "nr_inst_list.hpp"

#ifndef NR_INST_PART
#error "compile with -DNR_INST_PART=0..3"
#endif

namespace mapdn {

#define NR_CAT_(a, b) a##b
#define NR_CAT(a, b) NR_CAT_(a, b)
#define NR_ENTRY(w, l, hl, gl, res) {w, l, hl, gl, res, (const void*)k_nr_tree<w, l, hl, gl, res>},

extern const NrInst NR_CAT(nr_insts_, NR_INST_PART)[] = { NR_CAT(NR_INSTS_, NR_INST_PART)(NR_ENTRY) };
extern const int NR_CAT(nr_n_insts_, NR_INST_PART) = (int)(sizeof(NR_CAT(nr_insts_, NR_INST_PART)) / sizeof(NrInst));

#ifdef MAPDN_NR_STAMPS
// debug build only: the cycle stamps live in a per-part device array (no relocatable device code); launch_nr remembers the part
int NR_CAT(nr_debug_stamps_, NR_INST_PART)(unsigned long long* out, int n) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), (size_t)n * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
#endif

}  // namespace mapdn
