// Part of trg_engine.cpp (included inside its anonymous namespace): the host replay of expandGraph
// (trg.cpp:372-454) over chunks of GPU-evaluated samples and edges -- updateGraph's path and the fallback of the
// device-resident BFS (trg_engine_bfs.inc).
// ---- chunk pipeline ----------------------------------------------------------------------------
TrgStatus ensure_chunks(TrgEngine *e) {
  const int S = e->prm.sample_num;
  if (e->chunk_S == S && e->chunks[0].done) return TRG_OK;
  const size_t cmax = TrgEngine::CHUNK_MAX;
  const size_t slots = cmax * (size_t)std::max(S, 1);
  Chunk *all_chunks[TrgEngine::NCHUNK + 1];
  for (int i = 0; i < TrgEngine::NCHUNK; ++i) all_chunks[i] = &e->chunks[i];
  all_chunks[TrgEngine::NCHUNK] = &e->root_chunk;
  for (Chunk *cp : all_chunks) {
    Chunk &c = *cp;
    if (!c.done) {
      HIPCHK(e, hipEventCreateWithFlags(&c.done, hipEventDisableTiming));
      HIPCHK(e, hipEventCreate(&c.t0));
      HIPCHK(e, hipEventCreate(&c.t1));
      HIPCHK(e, hipEventCreate(&c.t2));
    }
    HIPCHK(e, alloc_pinned(c.in_blob, 6 * cmax));
    HIPCHK(e, alloc_pinned(c.out_blob, 2 * cmax + 6 * slots));
    HIPCHK(e, alloc_pinned(c.mt, 4 + 4 * (size_t)MAPTIE_CAP));
    if (c.mid_cap < slots) {
      if (c.d_mid) (void)hipFree(c.d_mid);
      c.d_mid = nullptr;
      HIPCHK(e, hipMalloc((void **)&c.d_mid, edge_mid_floats(slots) * sizeof(float)));
      c.mid_cap = slots;
    }
  }
  for (EdgeBatch &b : e->ebatches) {
    if (!b.done) {
      HIPCHK(e, hipEventCreateWithFlags(&b.done, hipEventDisableTiming));
      HIPCHK(e, hipEventCreate(&b.t0));
      HIPCHK(e, hipEventCreate(&b.t1));
    }
    HIPCHK(e, alloc_pinned(b.p1, (size_t)TrgEngine::EBATCH_MAX * 3));
    HIPCHK(e, alloc_pinned(b.p2, (size_t)TrgEngine::EBATCH_MAX * 3));
    HIPCHK(e, alloc_pinned(b.weight, (size_t)TrgEngine::EBATCH_MAX));
    HIPCHK(e, alloc_pinned(b.dist, (size_t)TrgEngine::EBATCH_MAX));
    HIPCHK(e, alloc_pinned(b.status, (size_t)TrgEngine::EBATCH_MAX));
    if (!b.d_mid)
      HIPCHK(e, hipMalloc((void **)&b.d_mid,
                          edge_mid_floats(TrgEngine::EBATCH_MAX) * sizeof(float)));
  }
  e->chunk_S = S;
  return TRG_OK;
}

TrgStatus submit_chunk(TrgEngine *e, Chunk &c, int first, int count) {
  const int S = e->prm.sample_num;
  c.first = first;
  c.count = count;
  c.carve(count, S);
  for (int i = 0; i < count; ++i) {
    const int id = e->queue[first + i];
    c.node_xy.h[2 * i] = e->nx[id];
    c.node_xy.h[2 * i + 1] = e->ny[id];
    c.node_xyz.h[3 * i] = e->nx[id];
    c.node_xyz.h[3 * i + 1] = e->ny[id];
    c.node_xyz.h[3 * i + 2] = e->nz[id];
    c.node_id.h[i] = id;
  }
  hipStream_t s = e->s_main;
  HIPCHK(e, hipMemcpyAsync(c.in_blob.d, c.in_blob.h, c.in_words * sizeof(uint32_t),
                           hipMemcpyHostToDevice, s));
  const QueryParams q = qparams(e);
  HIPCHK(e, hipMemsetAsync(c.mt.d, 0, sizeof(int), s));
  HIPCHK(e, hipEventRecord(c.t0, s));
  launch_sample_nodes(e->gmap.view, q, e->d_cos, e->d_sin, e->sampler.table_bits, e->sampler.seed,
                      e->epoch, c.node_xy.d, c.node_id.d, count, c.n_acc.d, c.n_draws.d, c.sx.d,
                      c.sy.d, c.sz.d, e->d_ctr, c.mt.d, (MapTieRec *)(c.mt.d + 4), s);
  HIPCHK(e, hipEventRecord(c.t1, s));
  launch_spec_edges(e->gmap.view, q, c.node_xyz.d, count, c.n_acc.d, c.sx.d, c.sy.d, c.sz.d,
                    c.d_mid, c.status.d, nullptr, c.weight.d, c.dist.d, e->d_ctr, s);
  HIPCHK(e, hipEventRecord(c.t2, s));
  HIPCHK(e, hipMemcpyAsync(c.out_blob.h, c.out_blob.d, c.out_words * sizeof(uint32_t),
                           hipMemcpyDeviceToHost, s));
  HIPCHK(e, hipMemcpyAsync(c.mt.h, c.mt.d, (4 + 4 * (size_t)MAPTIE_CAP) * sizeof(int),
                           hipMemcpyDeviceToHost, s));
  HIPCHK(e, hipEventRecord(c.done, s));
  c.in_flight = true;
  e->stats.launches_sample_kernel++;
  e->stats.launches_spec_kernel++;
  return TRG_OK;
}

// the same launch sequence for an explicit list of node ids (the roots of updateGraph's expansions)
TrgStatus submit_nodes(TrgEngine *e, Chunk &c, const int *ids, int count) {
  std::vector<int> saved;
  saved.swap(e->queue);
  e->queue.assign(ids, ids + count);
  const TrgStatus st = submit_chunk(e, c, 0, count);
  e->queue.swap(saved);
  return st;
}

TrgStatus wait_chunk(TrgEngine *e, Chunk &c) {
  if (!c.in_flight) return TRG_OK;
  auto t0 = Clock::now();
  HIPCHK(e, hipEventSynchronize(c.done));
  e->stats.ms_wait_gpu += ms_since(t0);
  float ms = 0;
  if (hipEventElapsedTime(&ms, c.t0, c.t1) == hipSuccess) e->stats.ms_sample_kernel += ms;
  if (hipEventElapsedTime(&ms, c.t1, c.t2) == hipSuccess) e->stats.ms_spec_kernel += ms;
  c.in_flight = false;
  // accepted samples whose elevation hung on a nearest-point tie: take the point the reference's
  // map tree returns, and re-evaluate the parent edge if that changed the sample's z
  const int n_mt = c.mt.h[0];
  if (n_mt > 0) {
    const int S = e->prm.sample_num;
    const MapTieRec *recs = (const MapTieRec *)(c.mt.h + 4);
    if (n_mt > MAPTIE_CAP) e->stats.map_nn_unresolved += (uint64_t)(n_mt - MAPTIE_CAP);
    for (int k = 0; k < std::min(n_mt, MAPTIE_CAP); ++k) {
      const MapTieRec &r = recs[k];
      float z = 0;
      bool found = false;
      TrgStatus st = map_nn_exact(e, e->gmap, r.qx, r.qy, &z, &found);
      if (st != TRG_OK) return st;
      if (!found || !(z != c.sz.h[r.slot])) continue;
      c.sz.h[r.slot] = z;
      const int qi = r.slot / S;
      const float p1[3] = {c.node_xyz.h[3 * qi], c.node_xyz.h[3 * qi + 1], c.node_xyz.h[3 * qi + 2]};
      const float p2[3] = {r.qx, r.qy, z};
      int32_t stt = 0;
      float w = 0, d = 0;
      st = edges_sync(e, e->gmap, p1, p2, 1, &stt, nullptr, &w, &d, true);
      if (st != TRG_OK) return st;
      c.status.h[r.slot] = stt;
      c.weight.h[r.slot] = w;
      c.dist.h[r.slot] = d;
    }
  }
  return TRG_OK;
}

TrgStatus collect_batch(TrgEngine *e, EdgeBatch &b) {
  if (!b.in_flight) return TRG_OK;
  auto t0 = Clock::now();
  HIPCHK(e, hipEventSynchronize(b.done));
  e->stats.ms_wait_gpu += ms_since(t0);
  float ms = 0;
  if (hipEventElapsedTime(&ms, b.t0, b.t1) == hipSuccess) e->stats.ms_edge_kernel += ms;
  for (int i = 0; i < b.count; ++i) {
    CallRec &c = e->calls[b.call_idx[i]];
    c.dist = b.dist.h[i];
    c.status = resolve_status(e, b.status.h[i], e->nz[c.n1], e->nz[c.n2], c.dist);
    c.weight = (c.status == EDGE_OK) ? b.weight.h[i] : 0.0f;
  }
  b.in_flight = false;
  b.count = 0;
  return TRG_OK;
}

// ship the pending deferred wireEdge evaluations (node -> already existing node) to the GPU
TrgStatus flush_pending(TrgEngine *e, bool all) {
  size_t pos = 0;
  while (e->pending_calls.size() - pos >= (all ? 1u : (size_t)TrgEngine::EBATCH_MAX)) {
    // find a free batch buffer, collecting the oldest if none
    EdgeBatch *b = nullptr;
    for (EdgeBatch &cand : e->ebatches)
      if (!cand.in_flight) {
        b = &cand;
        break;
      }
    if (!b) {
      TrgStatus st = collect_batch(e, e->ebatches[0]);
      if (st != TRG_OK) return st;
      // rotate so that [0] is again the oldest
      std::rotate(e->ebatches, e->ebatches + 1, e->ebatches + TrgEngine::NEBATCH);
      b = &e->ebatches[TrgEngine::NEBATCH - 1];
    }
    const int cnt = (int)std::min<size_t>(TrgEngine::EBATCH_MAX, e->pending_calls.size() - pos);
    b->call_idx.assign(e->pending_calls.begin() + pos, e->pending_calls.begin() + pos + cnt);
    for (int i = 0; i < cnt; ++i) {
      const CallRec &c = e->calls[b->call_idx[i]];
      b->p1.h[3 * i] = e->nx[c.n1];
      b->p1.h[3 * i + 1] = e->ny[c.n1];
      b->p1.h[3 * i + 2] = e->nz[c.n1];
      b->p2.h[3 * i] = e->nx[c.n2];
      b->p2.h[3 * i + 1] = e->ny[c.n2];
      b->p2.h[3 * i + 2] = e->nz[c.n2];
    }
    hipStream_t s = e->s_edge;
    HIPCHK(e, hipMemcpyAsync(b->p1.d, b->p1.h, (size_t)cnt * 3 * sizeof(float),
                             hipMemcpyHostToDevice, s));
    HIPCHK(e, hipMemcpyAsync(b->p2.d, b->p2.h, (size_t)cnt * 3 * sizeof(float),
                             hipMemcpyHostToDevice, s));
    HIPCHK(e, hipEventRecord(b->t0, s));
    launch_edges(e->gmap.view, qparams(e), b->p1.d, b->p2.d, cnt, b->d_mid, b->status.d, nullptr,
                 b->weight.d, b->dist.d, e->d_ctr, s);
    HIPCHK(e, hipEventRecord(b->t1, s));
    HIPCHK(e, hipMemcpyAsync(b->status.h, b->status.d, (size_t)cnt * sizeof(int),
                             hipMemcpyDeviceToHost, s));
    HIPCHK(e, hipMemcpyAsync(b->weight.h, b->weight.d, (size_t)cnt * sizeof(float),
                             hipMemcpyDeviceToHost, s));
    HIPCHK(e, hipMemcpyAsync(b->dist.h, b->dist.d, (size_t)cnt * sizeof(float),
                             hipMemcpyDeviceToHost, s));
    HIPCHK(e, hipEventRecord(b->done, s));
    b->in_flight = true;
    b->count = cnt;
    e->stats.launches_edge_kernel++;
    e->stats.edge_evals_gpu += cnt;
    pos += cnt;
  }
  e->pending_calls.erase(e->pending_calls.begin(), e->pending_calls.begin() + pos);
  if (all) {
    for (EdgeBatch &b : e->ebatches) {
      TrgStatus st = collect_batch(e, b);
      if (st != TRG_OK) return st;
    }
  }
  return TRG_OK;
}

inline void emit_deferred(TrgEngine *e, int n1, int n2) {
  if (n1 == n2) return;  // wireEdge returns at once (trg.cpp:255-257)
  e->calls.push_back(CallRec{n1, n2, -1, 0.0f, 0.0f});
  e->pending_calls.push_back((int)e->calls.size() - 1);
  e->stats.edge_calls++;
}

// Apply the logged wireEdge() calls in program order: the dedupe of trg.cpp:255-267 and the two
// push_backs of trg.cpp:365-368.  `from` = first call not yet applied.
void apply_calls(TrgEngine *e, size_t from) {
  e->edges.grow_nodes(e->nx.size());
  for (size_t i = from; i < e->calls.size(); ++i) {
    const CallRec &c = e->calls[i];
    if (c.n1 == c.n2) continue;
    if (e->edges.has(c.n1, c.n2) || e->edges.has(c.n2, c.n1)) continue;
    if (c.status != EDGE_OK) continue;
    e->edges.push(c.n1, c.n2, c.weight, c.dist);
    e->edges.push(c.n2, c.n1, c.weight, c.dist);
  }
}

// BFS expansion from the node `ref_id`, replaying trg.cpp:372-454 with GPU results.
// `applied` is the index of the first call not yet folded into e->edges; step 3's validity test
// needs edges of brand-new nodes only, which it derives locally.
// pre / pre_qi: GPU results of the root already fetched (entry pre_qi of chunk *pre, used by
// updateGraph, which fetches all its roots in bulk); the root then needs no round trip of its own.
TrgStatus expand_bfs(TrgEngine *e, int ref_id, Chunk *pre = nullptr, int pre_qi = 0) {
  TrgStatus st = ensure_chunks(e);
  if (st != TRG_OK) return st;
  const int S = e->prm.sample_num;
  const float r = e->prm.robot_size;
  e->queue.clear();
  e->queue.push_back(ref_id);
  size_t head = 0;        // next queue position to replay
  size_t submitted = pre ? 1 : 0;  // queue positions [0, submitted) have been shipped to the GPU
  int next_buf = 0;       // chunk buffers are used round-robin, so completion order == queue order
  std::deque<int> inflight;  // chunk buffer indices in submission order
  std::vector<int> range_hits;
  std::vector<float> s3_p1, s3_p2, s3_w, s3_d;
  std::vector<int32_t> s3_st;
  auto t_replay = Clock::now();
  double waited0 = e->stats.ms_wait_gpu;

  // Ship queue positions [submitted, submitted+cnt) in the next free buffer.  Buffers are used
  // round-robin and consumed in the same order, so the oldest in-flight chunk is always next.
  auto ship = [&](size_t cnt) -> TrgStatus {
    Chunk &c = e->chunks[next_buf];
    TrgStatus s2 = submit_chunk(e, c, (int)submitted, (int)cnt);
    if (s2 != TRG_OK) return s2;
    inflight.push_back(next_buf);
    next_buf = (next_buf + 1) % TrgEngine::NCHUNK;
    submitted += cnt;
    return TRG_OK;
  };

  Chunk *cur = nullptr;
  while (head < e->queue.size()) {
    // keep the GPU fed while the replay works: full-size chunks as soon as enough nodes are queued,
    // a small one only when the chunk being replayed is about to run dry
    for (;;) {
      const size_t avail = e->queue.size() - submitted;
      const int busy = (int)inflight.size() + (cur ? 1 : 0);
      if (avail == 0 || busy >= TrgEngine::NCHUNK) break;
      const size_t left = cur ? (size_t)(cur->first + cur->count) - head : 0;
      const bool starving = inflight.empty() && left <= 16;
      if (avail < 512 && !starving) break;
      st = ship(std::min<size_t>(avail, TrgEngine::CHUNK_MAX));
      if (st != TRG_OK) return st;
    }
    const bool use_pre = pre && head == 0;
    if (!use_pre && (!cur || (int)head >= cur->first + cur->count)) {
      cur = nullptr;
      if (inflight.empty()) return e->fail(TRG_ERR_DEVICE, "replay starved (internal error)");
      cur = &e->chunks[inflight.front()];
      inflight.pop_front();
      st = wait_chunk(e, *cur);
      if (st != TRG_OK) return st;
    }
    Chunk *const src = use_pre ? pre : cur;
    const int qi = use_pre ? pre_qi : (int)head - cur->first;
    const int node = e->queue[head];
    head++;
    e->stats.expanded_nodes++;
    const int n_acc = src->n_acc.h[qi];
    e->stats.trials += src->n_draws.h[qi];
    e->stats.samples += n_acc;
    e->stats.edge_evals_gpu += n_acc;

    for (int j = 0; j < n_acc; ++j) {
      const int slot = qi * S + j;
      const float sx = src->sx.h[slot], sy = src->sy.h[slot];
      // 1. nearest existing node (trg.cpp:408-417)
      const int ex = nearest_node(e, sx, sy);
      if (e->nstate[ex] == TRG_NODE_INVALID) continue;
      if (norm2f(e->nx[ex] - sx, e->ny[ex] - sy) < r) {
        emit_deferred(e, node, ex);
        continue;
      }
      // 2. new node (trg.cpp:420-426); its parent edge was evaluated speculatively on the GPU
      const int new_state = (ref_id == 0) ? TRG_NODE_VALID : TRG_NODE_FRONTIER;
      const float sz = src->sz.h[slot];
      const int nn = add_node_host(e, sx, sy, sz, new_state);
      const float dist = src->dist.h[slot];
      const int stt = resolve_status(e, src->status.h[slot], e->nz[node], sz, dist);
      const bool parent_ok = (stt == EDGE_OK);
      e->calls.push_back(
          CallRec{node, nn, stt, parent_ok ? src->weight.h[slot] : 0.0f, dist});
      e->stats.edge_calls++;
      bool has_edge = parent_ok;

      // 3. neighbour wiring (trg.cpp:429-444), only for configs like indoor.yaml
      if (e->step3) {
        kd_sync(e);
        e->kd.range(e->nx[nn], e->ny[nn], e->prm.expand_dist, range_hits);
        const size_t first_call = e->calls.size();
        for (int other : range_hits) {
          if (e->nstate[other] == TRG_NODE_INVALID) continue;
          emit_deferred(e, nn, other);
        }
        if (!parent_ok && e->calls.size() > first_call) {
          // the node's fate hangs on these edges: evaluate them now (synchronous round trip)
          const size_t m = e->calls.size() - first_call;
          s3_p1.resize(3 * m);
          s3_p2.resize(3 * m);
          s3_st.resize(m);
          s3_w.resize(m);
          s3_d.resize(m);
          for (size_t k = 0; k < m; ++k) {
            const CallRec &c = e->calls[first_call + k];
            s3_p1[3 * k] = e->nx[c.n1];
            s3_p1[3 * k + 1] = e->ny[c.n1];
            s3_p1[3 * k + 2] = e->nz[c.n1];
            s3_p2[3 * k] = e->nx[c.n2];
            s3_p2[3 * k + 1] = e->ny[c.n2];
            s3_p2[3 * k + 2] = e->nz[c.n2];
          }
          st = edges_sync(e, e->gmap, s3_p1.data(), s3_p2.data(), m, s3_st.data(), nullptr,
                          s3_w.data(), s3_d.data(), true);
          if (st != TRG_OK) return st;
          for (size_t k = 0; k < m; ++k) {
            CallRec &c = e->calls[first_call + k];
            c.status = s3_st[k];
            c.weight = s3_w[k];
            c.dist = s3_d[k];
            if (c.status == EDGE_OK) has_edge = true;
          }
          // they are resolved: take them off the pending list (they were appended last)
          e->pending_calls.resize(e->pending_calls.size() - m);
        }
      }

      // 4. (trg.cpp:447-451)
      if (!has_edge) {
        e->nstate[nn] = TRG_NODE_INVALID;
        e->stats.invalid_nodes++;
        continue;
      }
      e->queue.push_back(nn);
    }
    if ((int)e->pending_calls.size() >= TrgEngine::EBATCH_MAX) {
      st = flush_pending(e, false);
      if (st != TRG_OK) return st;
    }
  }
  e->stats.ms_replay_host += ms_since(t_replay) - (e->stats.ms_wait_gpu - waited0);
  return TRG_OK;
}

