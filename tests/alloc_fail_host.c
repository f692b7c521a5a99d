/* Test host (TEST INFRASTRUCTURE): a real allocation failure inside mapdn_create must come back as an error CODE, never as a C++
 * exception crossing the extern "C" boundary ("never throw across the boundary").
 * Builds a 3 000-bus radial feeder, caps the address space (RLIMIT_AS) just above what the process already holds so that the plan's
 * std::vector allocations fail, expects MAPDN_E_NOMEM + text from mapdn_create(device = -1), lifts the cap and creates again. */
#define _POSIX_C_SOURCE 200809L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/resource.h>

#include "mapdn.h"

static long vm_bytes(void) {
  long pages = 0;
  FILE* f = fopen("/proc/self/statm", "r");
  if (!f) return -1;
  if (fscanf(f, "%ld", &pages) != 1) pages = -1;
  fclose(f);
  return pages < 0 ? -1 : pages * 4096L;
}

int main(void) {
  const int nb = 3000, nl = nb - 1;
  double* vn = malloc(sizeof(double) * nb); int32_t* zone = malloc(sizeof(int32_t) * nb);
  int32_t* fb = malloc(sizeof(int32_t) * nl); int32_t* tb = malloc(sizeof(int32_t) * nl); int32_t* par = malloc(sizeof(int32_t) * nl);
  double* r = malloc(sizeof(double) * nl); double* x = malloc(sizeof(double) * nl); double* z = calloc(nl, sizeof(double));
  double* len = malloc(sizeof(double) * nl); uint8_t* on = malloc(nl); int32_t* lb = malloc(sizeof(int32_t) * nl);
  int32_t sgen_bus[2], sgen_zone[2] = {1, 2};
  if (!vn || !zone || !fb || !tb || !par || !r || !x || !z || !len || !on || !lb) return 10;
  for (int i = 0; i < nb; ++i) { vn[i] = 12.66; zone[i] = i == 0 ? 0 : (i < nb / 2 ? 1 : 2); }
  for (int i = 0; i < nl; ++i) { fb[i] = i; tb[i] = i + 1; par[i] = 1; r[i] = 1e-4; x[i] = 5e-5; len[i] = 1.0; on[i] = 1; lb[i] = i + 1; }
  sgen_bus[0] = nb / 4; sgen_bus[1] = 3 * (nb / 4);

  mapdn_netspec net;
  memset(&net, 0, sizeof net);
  net.n_bus = nb; net.bus_vn_kv = vn; net.bus_zone = zone;
  net.n_line = nl; net.line_from_bus = fb; net.line_to_bus = tb; net.line_r_ohm_per_km = r; net.line_x_ohm_per_km = x;
  net.line_c_nf_per_km = z; net.line_g_us_per_km = z; net.line_length_km = len; net.line_parallel = par; net.line_in_service = on;
  net.n_load = nl; net.load_bus = lb;
  net.n_sgen = 2; net.sgen_bus = sgen_bus; net.sgen_zone = sgen_zone;
  net.ext_grid_bus = 0; net.ext_grid_vm_pu = 1.0; net.sn_mva = 1.0; net.f_hz = 50.0;
  mapdn_env_config cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.barrier_type = MAPDN_BARRIER_BOWL; cfg.voltage_weight = 1.0; cfg.q_weight = 0.1; cfg.use_q_weight = 1;
  cfg.v_lower = 0.95; cfg.v_upper = 1.05; cfg.episode_limit = 240; cfg.action_low = -0.8; cfg.action_high = 0.8;
  cfg.reset_action = 1; cfg.state_space = MAPDN_SS_ALL;

  struct rlimit old, cap;
  if (getrlimit(RLIMIT_AS, &old) != 0) return 11;
  const long now = vm_bytes();
  if (now < 0) return 12;
  cap = old; cap.rlim_cur = (rlim_t)now + (rlim_t)(2L << 20);          /* 2 MiB of head room: far less than the plan of 3 000 buses (its dense Ybus export alone is 144 MB) */
  if (setrlimit(RLIMIT_AS, &cap) != 0) return 13;
  mapdn_handle* h = NULL;
  int rc = mapdn_create(&net, &cfg, 4, -1, &h);
  if (setrlimit(RLIMIT_AS, &old) != 0) return 14;
  printf("capped create -> %d (%s) handle %s\n", rc, mapdn_last_error(NULL), h ? "set" : "null");
  if (rc != MAPDN_E_NOMEM || h != NULL) return 1;
  /* the library is still usable: same call without the cap (it may be refused for its size — MAPDN_E_TOPOLOGY / INVALID — but by code) */
  rc = mapdn_create(&net, &cfg, 4, -1, &h);
  printf("uncapped create -> %d (%s)\n", rc, rc == MAPDN_OK ? "ok" : mapdn_last_error(NULL));
  if (rc == MAPDN_OK) mapdn_destroy(h);
  return rc == MAPDN_E_NOMEM || rc == MAPDN_E_INTERNAL ? 2 : 0;
}
