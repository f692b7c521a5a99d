/* Minimal C host of the C ABI (include/mapdn.h): builds a 5-bus radial feeder, creates a host-only handle
 * (device == -1: plan only, no GPU needed), and prints what any FFI binding would read back — dimensions,
 * the per-unit Ybus the library built and the elimination schedule of the Newton-Raphson kernel; then closes a
 * tie line (meshed net) and reads back the general sparse solver's program.
 * With a GPU, pass a device index instead of -1 and continue with mapdn_set_profiles / mapdn_reset /
 * mapdn_step as INTEGRATION.md shows.
 *
 *   gcc -std=c99 -Wall -Wextra -pedantic -Iinclude examples/c_abi_host.c -o c_abi_host \
 *       -Lmapdn_amd -lmapdn_hip -Wl,-rpath,$PWD/mapdn_amd
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mapdn.h"

int main(void) {
  /* bus 0 = slack; 0-1-2 trunk, 2-3 and 2-4 laterals; one PV at bus 3 (zone 1) and one at bus 4 (zone 2) */
  const double vn[5] = {12.66, 12.66, 12.66, 12.66, 12.66};
  const int32_t zone[5] = {0, 0, 0, 1, 2};
  const int32_t fb[4] = {0, 1, 2, 2}, tb[4] = {1, 2, 3, 4};
  const double r[4] = {0.09, 0.49, 0.37, 0.38}, x[4] = {0.05, 0.25, 0.19, 0.19}, c[4] = {0, 0, 0, 0}, g[4] = {0, 0, 0, 0};
  const double len[4] = {1, 1, 1, 1};
  const int32_t par[4] = {1, 1, 1, 1};
  const uint8_t in_service[4] = {1, 1, 1, 1};
  const int32_t load_bus[4] = {1, 2, 3, 4}, sgen_bus[2] = {3, 4}, sgen_zone[2] = {1, 2};

  mapdn_netspec net;
  memset(&net, 0, sizeof net);
  net.n_bus = 5; net.bus_vn_kv = vn; net.bus_zone = zone;
  net.n_line = 4; net.line_from_bus = fb; net.line_to_bus = tb; net.line_r_ohm_per_km = r; net.line_x_ohm_per_km = x;
  net.line_c_nf_per_km = c; net.line_g_us_per_km = g; net.line_length_km = len; net.line_parallel = par;
  net.line_in_service = in_service;
  net.n_load = 4; net.load_bus = load_bus;
  net.n_sgen = 2; net.sgen_bus = sgen_bus; net.sgen_zone = sgen_zone;
  net.ext_grid_bus = 0; net.ext_grid_vm_pu = 1.0; net.sn_mva = 1.0; net.f_hz = 50.0;

  mapdn_env_config cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.barrier_type = MAPDN_BARRIER_BOWL; cfg.voltage_weight = 1.0; cfg.q_weight = 0.1; cfg.use_q_weight = 1;
  cfg.v_lower = 0.95; cfg.v_upper = 1.05; cfg.episode_limit = 240; cfg.action_low = -0.8; cfg.action_high = 0.8;
  cfg.reset_action = 1; cfg.state_space = MAPDN_SS_ALL; cfg.seed = 0; cfg.env_id_offset = 0;

  mapdn_handle* h = NULL;
  int rc = mapdn_create(&net, &cfg, 8, -1, &h);
  if (rc != MAPDN_OK) { fprintf(stderr, "mapdn_create: %d %s\n", rc, mapdn_last_error(NULL)); return 1; }

  mapdn_dims_t d;
  if (mapdn_dims(h, &d) != MAPDN_OK) return 2;
  printf("library built from sources with %s\n", mapdn_build_info());
  printf("n_envs %d n_bus %d n_line %d n_load %d n_sgen %d n_agents %d obs_size %d state_size %d radial %d\n",
         d.n_envs, d.n_bus, d.n_line, d.n_load, d.n_sgen, d.n_agents, d.obs_size, d.state_size, d.is_radial);

  double* y = (double*)malloc(sizeof(double) * 2 * 25);
  if (mapdn_get_ybus_dense(h, y) != MAPDN_OK) return 3;
  printf("Y[1][1] = %.6f%+.6fj  Y[1][2] = %.6f%+.6fj\n", y[2 * 6], y[2 * 6 + 1], y[2 * 7], y[2 * 7 + 1]);
  free(y);

  int32_t n_rows = 0, rows[2 * 8], parent[4];
  if (mapdn_get_schedule(h, 2, &n_rows, NULL, NULL) != MAPDN_OK || n_rows > 8) return 4;
  if (mapdn_get_schedule(h, 2, &n_rows, rows, parent) != MAPDN_OK) return 5;
  printf("NR schedule for 2 workers: %d rows; parents", n_rows);
  for (int k = 0; k < 4; ++k) printf(" %d", parent[k]);
  printf("\n");

  /* a device entry point on a host-only handle fails cleanly with an error code and text */
  rc = mapdn_reset(h, NULL, 1, 3, NULL);
  printf("mapdn_reset on a host-only handle -> %d (%s)\n", rc, mapdn_last_error(h));
  mapdn_destroy(h);
  if (rc != MAPDN_E_STATE) return 6;

  /* close a tie line 3-4: the net is meshed now and takes the general sparse solver, whose elimination program (minimum
   * degree order, fill, list-scheduled 2x2 block operations) can be read back like the tree schedule */
  {
    const int32_t fb2[5] = {0, 1, 2, 2, 3}, tb2[5] = {1, 2, 3, 4, 4};
    const double r2[5] = {0.09, 0.49, 0.37, 0.38, 0.5}, x2[5] = {0.05, 0.25, 0.19, 0.19, 0.3}, z5[5] = {0, 0, 0, 0, 0};
    const double len2[5] = {1, 1, 1, 1, 1};
    const int32_t par2[5] = {1, 1, 1, 1, 1};
    const uint8_t on2[5] = {1, 1, 1, 1, 1};
    int32_t dims[6];
    net.n_line = 5; net.line_from_bus = fb2; net.line_to_bus = tb2; net.line_r_ohm_per_km = r2; net.line_x_ohm_per_km = x2;
    net.line_c_nf_per_km = z5; net.line_g_us_per_km = z5; net.line_length_km = len2; net.line_parallel = par2; net.line_in_service = on2;
    h = NULL;
    rc = mapdn_create(&net, &cfg, 8, -1, &h);
    if (rc != MAPDN_OK) { fprintf(stderr, "mapdn_create (meshed): %d %s\n", rc, mapdn_last_error(NULL)); return 7; }
    if (mapdn_dims(h, &d) != MAPDN_OK || d.is_radial != 0) return 8;
    if (mapdn_get_sparse_program(h, 8, dims, NULL, NULL, NULL) != MAPDN_OK) return 9;
    printf("meshed (tie 3-4 closed): radial %d; sparse program for 8 sub-lanes: %d block slots, %d fill, %d phases\n",
           d.is_radial, dims[0], dims[1], dims[2]);
    mapdn_destroy(h);
  }
  return 0;
}
