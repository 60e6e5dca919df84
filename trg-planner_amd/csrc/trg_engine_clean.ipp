// Part of trg_engine.cpp (included inside its anonymous namespace): cleanGraph on the host (trg.cpp:491-535,
// the update path; a build's cleanGraph runs on the device, k_fin_*), the CSR snapshot and the local graph
// (trg.cpp:211-231).
// ---- cleanGraph (trg.cpp:491-535) ---------------------------------------------------------------
void snapshot_csr(const TrgEngine *e, Csr &out) {
  const size_t V = e->nx.size();
  out.clear();
  out.xyz.resize(3 * V);
  out.state.resize(V);
  out.cid.resize(V);
  out.rowptr.resize(V + 1);
  out.rowptr[0] = 0;
  for (size_t i = 0; i < V; ++i) out.rowptr[i + 1] = out.rowptr[i] + (i < e->edges.deg.size() ? e->edges.deg[i] : 0);
  const size_t E = out.rowptr[V];
  out.col.resize(E);
  out.w.resize(E);
  out.dist.resize(E);
  parallel_ranges(V, [&](size_t i0, size_t i1) {
    for (size_t i = i0; i < i1; ++i) {
      out.xyz[3 * i] = e->nx[i];
      out.xyz[3 * i + 1] = e->ny[i];
      out.xyz[3 * i + 2] = e->nz[i];
      out.state[i] = e->nstate[i];
      out.cid[i] = e->ncid[i];
      int k = out.rowptr[i];
      if (i >= e->edges.head.size()) continue;
      for (int ed = e->edges.head[i]; ed >= 0; ed = e->edges.next[ed]) {
        out.col[k] = e->edges.dst[ed];
        out.w[k] = e->edges.w[ed];
        out.dist[k] = e->edges.dist[ed];
        ++k;
      }
    }
  });
}

void clean_graph(TrgEngine *e) {
  const bool trace = getenv("TRG_TIMING") != nullptr;
  const auto t_cg = Clock::now();
  auto lapc = [&](const char *what) {
    if (trace) fprintf(stderr, "[trg cleanGraph]   %-24s %8.3f ms\n", what, ms_since(t_cg));
  };
  const size_t V = e->nx.size();
  std::vector<int> old2new(V, 0);  // old2new[] default-constructs 0 in the reference too
  std::vector<int> keep_order;     // old ids in the order they receive new ids
  int new_id = 0;
  // new ids follow the iteration order of the reference's unordered_map (trg.cpp:497-504)
  std::vector<int> order;
  node_map_order(e, order);
  for (const int id : order) {
    if (e->nstate[id] == TRG_NODE_INVALID || e->edges.deg[id] < 1) continue;
    old2new[id] = new_id;
    keep_order.push_back(id);
    new_id++;
  }
  // trg.cpp:505-520 drops the edges of kept nodes that lead to a node it deletes.  A kept node has edges,
  // so the only deleted nodes an edge can lead to are Invalid ones: "is deleted" is the state test below
  // (no separate marking pass over the 700 k-entry edge pool).
  lapc("renumbering");
  const int Vn = new_id;
  std::vector<float> x2(Vn), y2(Vn), z2(Vn);
  std::vector<int> st2(Vn), cid2(Vn);
  // the surviving rows in bulk on several host threads: counts, offsets, then every row copied with its
  // targets renumbered (a 6.7 M-entry pool rebuilt push by push cost 65 ms per updateGraph at C3)
  std::vector<int> offs((size_t)Vn + 1, 0);
  parallel_ranges((size_t)Vn, [&](size_t k0, size_t k1) {
    for (size_t k = k0; k < k1; ++k) {
      const int old = keep_order[k];
      int n = 0;
      for (int ed = e->edges.head[old]; ed >= 0; ed = e->edges.next[ed]) n += e->nstate[e->edges.dst[ed]] != TRG_NODE_INVALID;
      offs[k + 1] = n;
    }
  });
  for (int k = 0; k < Vn; ++k) offs[k + 1] += offs[k];
  EdgePool ep;
  ep.alloc_rows(offs);
  parallel_ranges((size_t)Vn, [&](size_t k0, size_t k1) {
    for (size_t k = k0; k < k1; ++k) {
      const int old = keep_order[k];
      x2[k] = e->nx[old];
      y2[k] = e->ny[old];
      z2[k] = e->nz[old];
      st2[k] = e->nstate[old];
      cid2[k] = e->ncid[old];
      int pos = offs[k];
      for (int ed = e->edges.head[old]; ed >= 0; ed = e->edges.next[ed]) {
        const int d = e->edges.dst[ed];
        if (e->nstate[d] == TRG_NODE_INVALID) continue;
        ep.dst[pos] = old2new[d];
        ep.w[pos] = e->edges.w[ed];
        ep.dist[pos] = e->edges.dist[ed];
        ++pos;
      }
    }
    ep.link_rows(offs, k0, k1);
  });
  lapc("rows copied");
  e->last_new2old = keep_order;
  e->nx.swap(x2);
  e->ny.swap(y2);
  e->nz.swap(z2);
  e->nstate.swap(st2);
  e->ncid.swap(cid2);
  e->edges = std::move(ep);
  e->node_id = Vn;
  // new_nodes[new_id] = node for the dense new ids (trg.cpp:502), global_graph.nodes = new_nodes (:526:
  // bucket count, policy and element order of the source are taken over -- what moving it in leaves
  // behind); then the node tree is refilled in that map's iteration order
  if (e->real_map_stale) {
    MapOrderSim new_nodes;
    new_nodes.fill((size_t)Vn);
    e->nodes_sim.assign_from(new_nodes);
    e->nodes_sim.iteration_order(e->kd_insert_order);
  } else {
    std::unordered_map<int, int> new_nodes;
    for (int k = 0; k < Vn; ++k) new_nodes[k] = k;
    e->order_map = std::move(new_nodes);
    e->kd_insert_order.clear();
    for (auto &kv : e->order_map) e->kd_insert_order.push_back(kv.first);
  }
  e->kd_order_dirty = false;
  e->kd_valid = false;
  lapc("container replica");
  grid_rebuild(e);
  lapc("node grid");
  e->host_grid_valid = true;
  e->pool_valid = true;
}

void read_counters(TrgEngine *e) {
  std::vector<DeviceCounters> h(COUNTER_SHARDS);
  if (hipMemcpy(h.data(), e->d_ctr, COUNTER_SHARDS * sizeof(DeviceCounters),
                hipMemcpyDeviceToHost) == hipSuccess) {
    unsigned long long sh = 0, eh = 0, ph = 0, ties = 0;
    for (const DeviceCounters &c : h) {
      sh += c.sample_hits;
      eh += c.edge_hits;
      ph += c.spec_hits;
      ties += c.nn_ties;
    }
    e->stats.bytes_sample_kernel = 12ull * (sh + e->lv_hits_sample);
    e->stats.bytes_edge_kernel = 12ull * eh;
    e->stats.bytes_spec_kernel = 12ull * (ph + e->lv_hits_spec);
    e->stats.map_nn_ties += ties;
  }
}

// ---- local graph (trg.cpp:211-231) ---------------------------------------------------------------
// membership of the local graph (trg.cpp:211-231): n[i] != 0 iff a local-map point lies within
// robot_size / 2 of node i (a disc-emptiness probe); only nodes inside the local map's bounding box
// (grown by that radius) can have one
TrgStatus local_membership(TrgEngine *e, std::vector<int32_t> &n) {
  const size_t V = e->nx.size();
  n.assign(V, 0);
  if (!e->lmap.valid || V == 0) return TRG_OK;
  const float rr = (float)(e->prm.robot_size * 0.5) * 1.01f + 1e-4f;
  const float bx0 = e->lmap.bounds[0] - rr, by0 = e->lmap.bounds[1] - rr;
  const float bx1 = e->lmap.bounds[2] + rr, by1 = e->lmap.bounds[3] + rr;
  std::vector<int> cand;
  std::vector<float> xy;
  for (size_t i = 0; i < V; ++i)
    if (e->nx[i] >= bx0 && e->nx[i] <= bx1 && e->ny[i] >= by0 && e->ny[i] <= by1) {
      cand.push_back((int)i);
      xy.push_back(e->nx[i]);
      xy.push_back(e->ny[i]);
    }
  if (cand.empty()) return TRG_OK;
  std::vector<int32_t> nc(cand.size(), 0);
  TrgStatus st = collision_sync(e, e->lmap, 0.0f, xy.data(), cand.size(), nullptr, nullptr, nc.data(), e->s_aux,
                                (float)(e->prm.robot_size * 0.5));
  if (st != TRG_OK) return st;
  for (size_t k = 0; k < cand.size(); ++k) n[cand[k]] = nc[k];
  return TRG_OK;
}

// member: the membership flags if the caller already has them (updateGraph probes before its host-side
// cleanGraph, while the GPU is still awake: after ~10 ms without work the first launch takes ~2 ms)
TrgStatus set_local_graph(TrgEngine *e, const std::vector<int32_t> *member = nullptr) {
  e->local_nodes.clear();
  e->lkd.clear();
  const size_t V = e->nx.size();
  if (V == 0) {
    e->local_map.clear();
    return TRG_OK;
  }
  std::vector<int32_t> own;
  if (!member || member->size() != V) {
    TrgStatus st = local_membership(e, own);
    if (st != TRG_OK) return st;
    member = &own;
  }
  const std::vector<int32_t> &n = *member;
  e->local_map.clear();  // resetGraph("local"): clear() keeps the bucket array, as the reference's does
  std::vector<int> global_order;
  node_map_order(e, global_order);
  for (int id : global_order) {
    if (n[id] == 0) continue;
    e->local_map[id] = id;
    e->lkd.insert(e->nx[id], e->ny[id], id);
  }
  for (auto &kv : e->local_map) e->local_nodes.push_back(kv.first);
  return TRG_OK;
}

