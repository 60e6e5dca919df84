// Part of trg_engine.cpp (included at file scope): planSafePath / setGoal / refinePath on the CSR (host A*,
// trg.cpp:537-565, 603-730) and their C ABI entries.

namespace {

// positions in the node tree's insertion order (cleanGraph refills the tree in the node map's iteration
// order, trg.cpp:525-530): only built when an answer really hangs on the tree's shape
void ensure_kd_order_arrays(TrgEngine *e) {
  if (e->kdo_version == e->graph_version) return;
  materialize_kd_order(e);
  const size_t K = e->kd_insert_order.size();
  e->kdo_x.resize(K);
  e->kdo_y.resize(K);
  e->kdo_index.assign(e->nx.size(), -1);
  for (size_t k = 0; k < K; ++k) {
    const int id = e->kd_insert_order[k];
    e->kdo_x[k] = e->nx[id];
    e->kdo_y[k] = e->ny[id];
    e->kdo_index[id] = (int)k;
  }
  e->kdo_version = e->graph_version;
}

// kd_nearest2 on node_tree (trg.cpp:615): the grid answers; an exact fp32 distance tie goes to the
// tree-order argument of kd_tie_winner (no tree is built)
int plan_nearest_node(TrgEngine *e, float qx, float qy, std::vector<int> &tied) {
  bool tie = false;
  int s = e->grid.nearest(qx, qy, &tie);
  if (!tie || s < 0) return s;
  e->stats.nn_ties++;
  e->grid.tied_set(qx, qy, e->grid.dist2(s, qx, qy), tied);
  ensure_kd_order_arrays(e);
  for (int &t : tied) t = e->kdo_index[t];
  std::sort(tied.begin(), tied.end());
  const int w = kd_tie_winner(e->kdo_x.data(), e->kdo_y.data(), (int)e->kdo_x.size(), qx, qy, tied);
  return e->kd_insert_order[w];
}

// First item of kd_nearest_range2(node_tree, goal, robot_size) (trg.cpp:544-546), or -1 when the set is
// empty: the hit set comes from the grid; with several hits the reference takes the one its walk reaches
// LAST (the result list is filled at the head).  A hit within rounding of the radius sends the question
// to the tree replica.
int plan_first_range_hit(TrgEngine *e, float qx, float qy, float r, std::vector<int> &hits) {
  bool doubt = false;
  e->grid.range_set(qx, qy, r, hits, &doubt);
  if (doubt) {
    kd_sync(e);
    e->kd.range(qx, qy, r, hits);
    return hits.empty() ? -1 : hits[0];
  }
  if (hits.empty()) return -1;
  if (hits.size() == 1) return hits[0];
  ensure_kd_order_arrays(e);
  const int K = (int)e->kdo_x.size();
  int last = e->kdo_index[hits[0]];
  for (size_t i = 1; i < hits.size(); ++i) {
    const int h = e->kdo_index[hits[i]];
    if (kd_range_first_of_two(e->kdo_x.data(), e->kdo_y.data(), K, qx, qy, last, h) == last) last = h;
  }
  return e->kd_insert_order[last];
}

// setGoal (trg.cpp:537-565) without side effects: the goal node and whether it lies within robot_size
void plan_goal_node(TrgEngine *e, PlanScratch &ps, const float goal_xyz[3], int *goal, bool *known) {
  const int hit = plan_first_range_hit(e, goal_xyz[0], goal_xyz[1], e->prm.robot_size, ps.hits);
  if (hit >= 0) {
    *goal = hit;
    *known = true;
    return;
  }
  // nearest node by the float norm, first in the node map's iteration order among equals (trg.cpp:549-557)
  float min_dist = std::numeric_limits<float>::max();
  int g = -1;
  std::vector<int> map_order;
  node_map_order(e, map_order);
  for (int id : map_order) {
    const float d = norm2f(e->nx[id] - goal_xyz[0], e->ny[id] - goal_xyz[1]);
    if (d < min_dist) {
      min_dist = d;
      g = id;
    }
  }
  *goal = g;
  *known = false;
}

// planSafePath (trg.cpp:603-690) on the CSR of the global graph: a row's entries are the node's edges in
// the reference's push order, the heap is std::push_heap / std::pop_heap with the reference's comparator
// (f_cost greater-than), so equal-cost ties fall exactly as in the reference's std::priority_queue.
TrgStatus plan_on_csr(TrgEngine *e, PlanScratch &ps, int start, int goal, float *path_xyz, int32_t max_points,
                      TrgPathInfo *info) {
  const Csr &G = e->csr_global;
  const size_t V = e->nx.size();
  const int32_t *rowptr = G.rowptr.data(), *col = G.col.data();
  const float *ew = G.w.data(), *ed = G.dist.data();
  const float *nx = e->nx.data(), *ny = e->ny.data();
  const int *nstate = e->nstate.data();
  ps.begin(V);
  const uint32_t gen = ps.gen;
  std::vector<PlanScratch::Opt> &pool = ps.pool;
  std::vector<int> &heap = ps.heap;
  auto cmp = [&pool](int a, int b) { return pool[a].f > pool[b].f; };
  const float gx = nx[goal], gy = ny[goal];
  const double sf = e->prm.safety_factor;

  info->direct_dist = norm2f(gx - nx[start], gy - ny[start]);
  {
    const double g_cost = 0.0;
    const double f_cost = g_cost + info->direct_dist;
    pool.push_back(PlanScratch::Opt{start, -1, (float)f_cost, (float)g_cost});
    heap.push_back(0);
    ps.open_gen[start] = gen;
    ps.open_idx[start] = 0;
  }
  while (!heap.empty()) {
    std::pop_heap(heap.begin(), heap.end(), cmp);
    const int oi = heap.back();
    heap.pop_back();
    const PlanScratch::Opt cur = pool[oi];
    ps.open_gen[cur.id] = 0;  // open_check.erase(current id)

    if (cur.id == goal) {
      std::vector<int> &chain = ps.chain;
      chain.clear();
      float sum_dist = 0.0, sum_weight = 0.0;
      for (int node = oi; node >= 0; node = pool[node].parent) {
        const PlanScratch::Opt &o = pool[node];
        if (o.parent >= 0) {
          const int pid = pool[o.parent].id;
          for (int k = rowptr[o.id]; k < rowptr[o.id + 1]; ++k)
            if (col[k] == pid) {
              sum_dist += ed[k];
              sum_weight += ew[k];
              break;
            }
        }
        chain.push_back(o.id);
      }
      const float avg_weight = sum_weight / chain.size();
      std::reverse(chain.begin(), chain.end());
      info->path_length = sum_dist;
      info->avg_risk = avg_weight;
      info->num_points = (int32_t)chain.size();
      if (path_xyz) {
        const int m = std::min<int>((int)chain.size(), max_points);
        for (int i = 0; i < m; ++i) {
          path_xyz[3 * i] = nx[chain[i]];
          path_xyz[3 * i + 1] = ny[chain[i]];
          path_xyz[3 * i + 2] = e->nz[chain[i]];
        }
      }
      return TRG_OK;
    }

    ps.close_gen[cur.id] = gen;
    for (int k = rowptr[cur.id]; k < rowptr[cur.id + 1]; ++k) {
      const int dst = col[k];
      if (dst < 0 || dst >= (int)V) continue;
      if (ps.close_gen[dst] == gen || nstate[dst] == TRG_NODE_INVALID) continue;
      const double next_g = cur.g + (sf * ew[k] + 1) * ed[k];
      const double next_f = next_g + norm2f(gx - nx[dst], gy - ny[dst]);
      pool.push_back(PlanScratch::Opt{dst, oi, (float)next_f, (float)next_g});
      const int ni = (int)pool.size() - 1;
      if (ps.open_gen[dst] != gen || pool[ni].g < pool[ps.open_idx[dst]].g) {
        heap.push_back(ni);
        std::push_heap(heap.begin(), heap.end(), cmp);
        ps.open_gen[dst] = gen;
        ps.open_idx[dst] = ni;
      }
    }
  }
  return TRG_ERR_NOT_FOUND;
}

// common head of plan / plan_batch: graph present, CSR rows and node grid current
TrgStatus plan_prepare(TrgEngine *e) {
  const size_t V = e->nx.size();
  if (V == 0) return e->fail(TRG_ERR_NO_GRAPH, "graph is empty");
  const Csr &G = e->csr_global;
  if (G.rowptr.size() != V + 1 || G.state.size() != V) {
    if (!e->pool_valid) return e->fail(TRG_ERR_NO_GRAPH, "no CSR of the current graph");
    snapshot_csr(e, e->csr_global);
  }
  ensure_host_grid(e);
  if (!e->plan_scratch) e->plan_scratch = new PlanScratch();
  return TRG_OK;
}

}  // namespace

extern "C" {

TrgStatus trg_engine_plan(TrgEngine *e, const float start_xy[2], const float goal_xyz[3],
                          float *path_xyz, int32_t max_points, TrgPathInfo *info) {
  if (!e || !start_xy || !goal_xyz || !info) return TRG_ERR_INVALID_ARG;
  info->direct_dist = info->path_length = info->avg_risk = 0.0f;
  info->num_points = 0;
  TrgStatus st = plan_prepare(e);
  if (st != TRG_OK) return st;
  PlanScratch &ps = *e->plan_scratch;
  // setGoal
  e->goal_pose2d[0] = goal_xyz[0];
  e->goal_pose2d[1] = goal_xyz[1];
  plan_goal_node(e, ps, goal_xyz, &e->goal_node, &e->goal_known);
  const int start = plan_nearest_node(e, start_xy[0], start_xy[1], ps.tied);
  st = plan_on_csr(e, ps, start, e->goal_node, path_xyz, max_points, info);
  if (st == TRG_ERR_NOT_FOUND) return e->fail(TRG_ERR_NOT_FOUND, "no path");
  return st;
}

// m consecutive planSafePath calls.  The searches are independent and only read the graph: the start /
// goal nodes are looked up in call order (the goal state the last call leaves is the reference's), the
// searches themselves run on up to 8 host threads, each with its own scratch.
TrgStatus trg_engine_plan_batch(TrgEngine *e, const float *starts_xy, const float *goals_xyz,
                                size_t m, float *path_xyz, int32_t path_cap, int32_t *offsets,
                                TrgPathInfo *infos) {
  if (!e || !offsets || (m && (!starts_xy || !goals_xyz || !infos)))
    return TRG_ERR_INVALID_ARG;
  if (path_cap < 0 || (path_cap > 0 && !path_xyz)) return e->fail(TRG_ERR_INVALID_ARG, "path buffer");
  offsets[0] = 0;
  if (m == 0) return TRG_OK;
  TrgStatus st = plan_prepare(e);
  if (st != TRG_OK) return st;
  PlanScratch &ps0 = *e->plan_scratch;
  std::vector<int> starts(m), goals(m);
  for (size_t k = 0; k < m; ++k) {
    infos[k].direct_dist = infos[k].path_length = infos[k].avg_risk = 0.0f;
    infos[k].num_points = 0;
    e->goal_pose2d[0] = goals_xyz[3 * k];
    e->goal_pose2d[1] = goals_xyz[3 * k + 1];
    plan_goal_node(e, ps0, goals_xyz + 3 * k, &e->goal_node, &e->goal_known);
    goals[k] = e->goal_node;
    starts[k] = plan_nearest_node(e, starts_xy[2 * k], starts_xy[2 * k + 1], ps0.tied);
  }
  std::vector<std::vector<float>> paths(m);
  std::vector<TrgStatus> sts(m, TRG_OK);
  const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
  const size_t nthr = std::min<size_t>(std::min<size_t>(m, 8), hw);
  auto work = [&](size_t t, PlanScratch &ps) {
    for (size_t k = t; k < m; k += nthr) {
      TrgPathInfo probe;
      probe.direct_dist = probe.path_length = probe.avg_risk = 0.0f;
      probe.num_points = 0;
      sts[k] = plan_on_csr(e, ps, starts[k], goals[k], nullptr, 0, &probe);
      if (sts[k] == TRG_OK) {
        paths[k].resize(3 * (size_t)probe.num_points);
        for (int i = 0; i < probe.num_points; ++i) {
          const int id = ps.chain[i];
          paths[k][3 * i] = e->nx[id];
          paths[k][3 * i + 1] = e->ny[id];
          paths[k][3 * i + 2] = e->nz[id];
        }
      }
      infos[k] = probe;
    }
  };
  if (nthr <= 1) {
    work(0, ps0);
  } else {
    std::vector<PlanScratch> extra(nthr - 1);
    std::vector<std::thread> thr;
    for (size_t t = 1; t < nthr; ++t) thr.emplace_back(work, t, std::ref(extra[t - 1]));
    work(0, ps0);
    for (auto &th : thr) th.join();
  }
  int32_t used = 0;
  for (size_t k = 0; k < m; ++k) {
    if (sts[k] != TRG_OK && sts[k] != TRG_ERR_NOT_FOUND) return e->fail(sts[k], "plan_batch");
    if (sts[k] == TRG_ERR_NOT_FOUND) infos[k].num_points = 0;
    const int32_t room = path_cap - used;
    const int32_t take = std::min<int32_t>(infos[k].num_points, room);
    if (take > 0) memcpy(path_xyz + 3 * (size_t)used, paths[k].data(), 3 * (size_t)take * sizeof(float));
    used += std::max<int32_t>(take, 0);
    offsets[k + 1] = used;
  }
  return TRG_OK;
}

// reference: TRG::checkReadched (sic) trg.cpp:567-574 and TRG::checkReplan trg.cpp:576-601
int32_t trg_engine_check_reached(TrgEngine *e, const float pos_xy[2]) {
  if (!e || !pos_xy) return 0;
  const float dist = norm2f(e->goal_pose2d[0] - pos_xy[0], e->goal_pose2d[1] - pos_xy[1]);
  return dist < e->prm.goal_tolerance ? 1 : 0;
}

int32_t trg_engine_check_replan(TrgEngine *e, const float pos_xy[2], const float *path_xyz,
                                int32_t n_path) {
  if (!e || !pos_xy) return 0;
  if (e->goal_node < 0 || e->goal_node >= (int)e->nx.size()) return 0;
  const int g = e->goal_node;
  const float dist2subgoal = norm2f(e->nx[g] - pos_xy[0], e->ny[g] - pos_xy[1]);
  if (!e->goal_known && dist2subgoal < e->prm.goal_tolerance) return 1;
  if (!e->goal_known && e->nstate[g] != TRG_NODE_FRONTIER) return 1;
  ensure_host_grid(e);
  std::vector<int> hits;
  for (int i = 0; i < n_path; ++i) {  // (the node grid answers; the tree replica only within rounding of the radius)
    int w = e->grid.within(path_xyz[3 * i], path_xyz[3 * i + 1], e->prm.robot_size);
    if (w < 0) {
      kd_sync(e);
      e->kd.range(path_xyz[3 * i], path_xyz[3 * i + 1], e->prm.robot_size, hits);
      w = hits.empty() ? 0 : 1;
    }
    if (!w) return 1;
  }
  return 0;
}

int32_t trg_engine_refine_path(const float *in_xyz, int32_t n_in, float *out_xyz, int32_t max_out) {
  if (!in_xyz || n_in <= 0) return 0;
  // point_between == 1: p0,p1,p1,p2,p2,...  then a 3-tap mean, last point passed through
  std::vector<float> dense;
  for (int i = 0; i + 1 < n_in; ++i) {
    dense.insert(dense.end(), in_xyz + 3 * i, in_xyz + 3 * i + 3);
    dense.insert(dense.end(), in_xyz + 3 * i + 3, in_xyz + 3 * i + 6);
  }
  const int nd = (int)(dense.size() / 3);
  int written = 0;
  for (int i = 0; i < nd; ++i) {
    float o[3];
    if (i == nd - 1) {
      o[0] = dense[3 * i];
      o[1] = dense[3 * i + 1];
      o[2] = dense[3 * i + 2];
    } else {
      float sum[3] = {0.0f, 0.0f, 0.0f};
      int cnt = 0;
      for (int j = i - 1; j < i + 2; ++j) {
        if (j < 0 || j >= nd) continue;
        for (int k = 0; k < 3; ++k) sum[k] += dense[3 * j + k];
        cnt++;
      }
      for (int k = 0; k < 3; ++k) o[k] = sum[k] / cnt;
    }
    if (out_xyz && written < max_out) {
      out_xyz[3 * written] = o[0];
      out_xyz[3 * written + 1] = o[1];
      out_xyz[3 * written + 2] = o[2];
    }
    written++;
  }
  return written;
}

}  // extern "C"
