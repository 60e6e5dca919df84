/* Plain C consumer of include/trg_engine.h: what a non-Python binding (cgo, JNI, a ROS node ...)
 * would do.  Builds a tiny terrain, a graph, asks for a path; exit code 0 on success. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/trg_engine.h"

int main(void) {
  const int nx = 120, ny = 120;
  float *xyz = (float *)malloc(sizeof(float) * 3 * nx * ny);
  unsigned s = 12345u;
  for (int j = 0; j < ny; ++j)
    for (int i = 0; i < nx; ++i) {
      float *p = xyz + 3 * (j * nx + i);
      s = s * 1664525u + 1013904223u;
      p[0] = 0.1f * i + 0.02f * ((s >> 8) & 0xFFFF) / 65536.0f;
      s = s * 1664525u + 1013904223u;
      p[1] = 0.1f * j + 0.02f * ((s >> 8) & 0xFFFF) / 65536.0f;
      p[2] = 0.3f * sinf(0.5f * p[0]) * cosf(0.4f * p[1]);
    }
  TrgParams prm = {0, 0.6f, 0.3f, 7, 0.16f, 0.1f, 0.5f, 3.0f, 0.8f};
  TrgEngine *e = NULL;
  if (trg_engine_create(&prm, 0, &e) != TRG_OK) {
    fprintf(stderr, "create: %s\n", trg_engine_last_error(e));
    return 1;
  }
  if (trg_engine_set_global_map(e, xyz, (size_t)nx * ny, 3) != TRG_OK) return 2;
  const float start[3] = {6.0f, 6.0f, 0.0f};
  TrgSampler smp = {42u, 16};
  if (trg_engine_init_graph(e, start, &smp) != TRG_OK) {
    fprintf(stderr, "init_graph: %s\n", trg_engine_last_error(e));
    return 3;
  }
  TrgCsrView g;
  if (trg_engine_export_csr(e, TRG_KIND_GLOBAL, &g) != TRG_OK) return 4;
  if (g.num_nodes < 100 || g.num_edges < g.num_nodes) return 5;
  for (int i = 0; i < g.num_nodes; ++i)
    if (g.rowptr[i + 1] <= g.rowptr[i] || g.node_state[i] == TRG_NODE_INVALID) return 6;
  const float s2[2] = {2.0f, 2.0f}, goal[3] = {10.0f, 9.5f, 0.0f};
  float path[3 * 4096];
  TrgPathInfo info;
  if (trg_engine_plan(e, s2, goal, path, 4096, &info) != TRG_OK || info.num_points < 2) return 7;
  /* the native stitch exchange as a single-tile "tiling" of one rank: the engine's own RCCL communicator
   * (unique id drawn here; a multi-process consumer hands the 128 bytes to its peers), all-gathers, assembly */
  {
    unsigned char id[TRG_COMM_ID_BYTES];
    const float core[4] = {-1.0f, -1.0f, 13.0f, 13.0f};
    int32_t nb = -1, nc = -1;
    TrgCsrView sg;
    if (trg_engine_comm_unique_id(e, id) != TRG_OK || trg_engine_comm_init(e, id, 1, 0) != TRG_OK) {
      fprintf(stderr, "comm: %s\n", trg_engine_last_error(e));
      return 8;
    }
    if (trg_engine_stitch_exchange(e, core, 1, 1, &nb, &nc) != TRG_OK) {
      fprintf(stderr, "stitch_exchange: %s\n", trg_engine_last_error(e));
      return 9;
    }
    if (nb != 0 || nc != 0) return 10; /* one tile: no shared border */
    if (trg_engine_export_csr(e, TRG_KIND_STITCHED, &sg) != TRG_OK) return 11;
    if (sg.num_nodes != g.num_nodes || sg.num_edges != g.num_edges) return 12;
    trg_engine_comm_destroy(e);
  }
  TrgStats st;
  trg_engine_get_stats(e, &st);
  printf("ok arch=%s V=%d E=%d path=%d len=%.3f device_bfs=%llu levels=%llu\n",
         trg_engine_device_arch(e), g.num_nodes, g.num_edges, info.num_points, info.path_length,
         (unsigned long long)st.used_device_bfs, (unsigned long long)st.bfs_levels);
  trg_engine_destroy(e);
  free(xyz);
  return 0;
}
