// Part of trg_engine.cpp (included inside its anonymous namespace): the synchronous probes (isCollision,
// nearest map point, single edges) and the exact nearest-map-point tie-break against the reference's kd-tree order.
// ---- synchronous probes ------------------------------------------------------------------------
TrgStatus ensure_sync_scratch(TrgEngine *e, size_t m) {
  if (e->sy_cap >= m) return TRG_OK;
  size_t cap = std::max<size_t>(m, 1024);
  HIPCHK(e, alloc_pinned(e->sy_in, cap * 3));
  HIPCHK(e, alloc_pinned(e->sy_in2, cap * 3));
  HIPCHK(e, alloc_pinned(e->sy_f0, cap));
  HIPCHK(e, alloc_pinned(e->sy_f1, cap));
  HIPCHK(e, alloc_pinned(e->sy_i0, cap));
  HIPCHK(e, alloc_pinned(e->sy_i1, cap));
  HIPCHK(e, alloc_pinned(e->sy_i2, cap));
  if (e->sy_mid) (void)hipFree(e->sy_mid);
  e->sy_mid = nullptr;
  HIPCHK(e, hipMalloc((void **)&e->sy_mid, edge_mid_floats(cap) * sizeof(float)));
  e->sy_cap = cap;
  return TRG_OK;
}

DevMap *pick_map(TrgEngine *e, TrgKind k) { return k == TRG_KIND_LOCAL ? &e->lmap : &e->gmap; }

// strm: the stream the probe runs in (the main stream may still hold look-ahead work of a finished replay that
// nobody needs to wait for: the maps are read-only here)
// radius > 0: probe discs of that radius instead of robot_size (setLocalGraph / isFrontier ask for
// robot_size / 2, trg.cpp:214, 791) -- passed in the query parameters, the engine's own stay untouched
TrgStatus collision_sync(TrgEngine *e, DevMap &m, float threshold, const float *xy, size_t cnt,
                         int32_t *flag, int32_t *c_out, int32_t *n_out, hipStream_t strm = nullptr,
                         float radius = 0.0f) {
  if (!strm) strm = e->s_main;
  QueryParams qp = qparams(e);
  if (radius > 0.0f) qp.robot_size = radius;
  if (!m.valid) {
    // empty map: kd_nearest_range on an empty tree returns no hits -> collision (trg.cpp:749-752)
    for (size_t i = 0; i < cnt; ++i) {
      if (flag) flag[i] = 1;
      if (c_out) c_out[i] = 0;
      if (n_out) n_out[i] = 0;
    }
    return TRG_OK;
  }
  const size_t B = 1 << 20;
  for (size_t off = 0; off < cnt; off += B) {
    const size_t m_ = std::min(B, cnt - off);
    TrgStatus st = ensure_sync_scratch(e, m_);
    if (st != TRG_OK) return st;
    memcpy(e->sy_in.h, xy + 2 * off, m_ * 2 * sizeof(float));
    HIPCHK(e, hipMemcpyAsync(e->sy_in.d, e->sy_in.h, m_ * 2 * sizeof(float), hipMemcpyHostToDevice,
                             strm));
    launch_probe_collision(m.view, qp, threshold, e->sy_in.d, (int)m_, e->sy_i0.d,
                           e->sy_i1.d, e->sy_i2.d, e->d_ctr, strm);
    HIPCHK(e, hipMemcpyAsync(e->sy_i0.h, e->sy_i0.d, m_ * sizeof(int), hipMemcpyDeviceToHost,
                             strm));
    HIPCHK(e, hipMemcpyAsync(e->sy_i1.h, e->sy_i1.d, m_ * sizeof(int), hipMemcpyDeviceToHost,
                             strm));
    HIPCHK(e, hipMemcpyAsync(e->sy_i2.h, e->sy_i2.d, m_ * sizeof(int), hipMemcpyDeviceToHost,
                             strm));
    HIPCHK(e, hipStreamSynchronize(strm));
    HIPCHK(e, hipGetLastError());
    if (flag) memcpy(flag + off, e->sy_i0.h, m_ * sizeof(int));
    if (c_out) memcpy(c_out + off, e->sy_i1.h, m_ * sizeof(int));
    if (n_out) memcpy(n_out + off, e->sy_i2.h, m_ * sizeof(int));
  }
  e->stats.sync_batches++;
  return TRG_OK;
}

// ---- exact nearest-map-point tie-break ------------------------------------------------------------
// addNode takes the z of kd_nearest's result (trg.cpp:244-247).  When several map points are at the
// same minimal fp32 distance, kd_nearest returns the one it visits first (strict `<`, kdtree.c:343;
// the root is the initial best, kdtree.c:393-396), which depends on the shape of the insertion-built
// map tree.  The tree is never built here: as in kd_first_of_two (host_index.h) the visiting order
// of two tied points is decided at their lowest common ancestor, and the common path is followed
// by asking the GPU for "the point with the smallest original index inside this half-open region,
// inserted after the current ancestor" -- which is exactly the root of that subtree, because
// kd_insert appends in cloud order and sends `<` to the left (kdtree.c:179-198).
struct TiePoint {
  int perm;
  float x, y;
};

TrgStatus ensure_tie_scratch(TrgEngine *e) {
  if (e->mt_set_d) return TRG_OK;
  HIPCHK(e, hipMalloc((void **)&e->mt_set_d, sizeof(MapTieSet)));
  HIPCHK(e, hipMalloc((void **)&e->mt_walk_d, sizeof(MapTieWalk)));
  HIPCHK(e, hipHostMalloc((void **)&e->mt_set_h, sizeof(MapTieSet), hipHostMallocDefault));
  HIPCHK(e, hipHostMalloc((void **)&e->mt_walk_h, sizeof(MapTieWalk), hipHostMallocDefault));
  return TRG_OK;
}

// which of the tied points A, B the nearest-neighbour search for q visits first: 0 = A, 1 = B.  The
// walk (region scan + decision per tree level, ~30-50 levels on a 10 M-point map) runs on the device
// without the host in between: the first steps (regions of millions of points) one grid-wide kernel
// each, enqueued blindly, the rest inside a single workgroup; steps after the decision return at once.
constexpr int MAP_TOP_POINTS = 8192;  // points of the host-side top of the map tree

// The first MAP_TOP_POINTS points of the cloud, inserted like kd_insert does (kdtree.c:179-198: `<` goes
// left, the axis alternates with the depth): the top of the reference's map tree, node k = cloud point k.
static void insert_map_top(DevMap &m, int M) {
  m.top_left.assign(M, -1);
  m.top_right.assign(M, -1);
  for (int k = 1; k < M; ++k) {
    const float px = m.top_xy[2 * (size_t)k], py = m.top_xy[2 * (size_t)k + 1];
    int cur = 0, axis = 0;
    for (;;) {
      const float split = axis ? m.top_xy[2 * (size_t)cur + 1] : m.top_xy[2 * (size_t)cur];
      int &child = ((axis ? py : px) < split) ? m.top_left[cur] : m.top_right[cur];
      if (child < 0) {
        child = k;
        break;
      }
      cur = child;
      axis ^= 1;
    }
  }
  m.top_m = M;
}

TrgStatus ensure_map_top(TrgEngine *e, DevMap &m) {
  m.top_wait();  // (the global map's top is prepared beside the build)
  if (m.top_m > 0) return TRG_OK;
  const int M = (int)std::min<size_t>(m.n, MAP_TOP_POINTS);
  float *d_xy = nullptr;
  HIPCHK(e, hipMalloc((void **)&d_xy, (size_t)M * 2 * sizeof(float)));
  launch_collect_first(m.view, M, d_xy, e->s_aux);
  m.top_xy.resize((size_t)M * 2);
  hipError_t he = hipMemcpyAsync(m.top_xy.data(), d_xy, (size_t)M * 2 * sizeof(float), hipMemcpyDeviceToHost, e->s_aux);
  if (he == hipSuccess) he = hipStreamSynchronize(e->s_aux);
  (void)hipFree(d_xy);
  if (he != hipSuccess) return e->fail(TRG_ERR_DEVICE, std::string("map top: ") + hipGetErrorString(he));
  insert_map_top(m, M);
  return TRG_OK;
}

// The same beside the build: the tie-breaking scratch is allocated, the first points are fetched on the aux
// stream, and a helper thread waits for them and inserts them while the graph is being built -- the first
// nearest-point tie of a build otherwise paid ~0.9 ms for all of this inside the level loop.  Failures are
// silent here: ensure_map_top then does the work on demand.
void start_map_top(TrgEngine *e, DevMap &m) {
  m.top_wait();
  m.top_m = 0;
  if (m.n == 0) return;
  if (ensure_tie_scratch(e) != TRG_OK) return;
  const int M = (int)std::min<size_t>(m.n, MAP_TOP_POINTS);
  if (!e->top_xy_d && hipMalloc((void **)&e->top_xy_d, (size_t)MAP_TOP_POINTS * 2 * sizeof(float)) != hipSuccess) return;
  if (!e->top_xy_h &&
      hipHostMalloc((void **)&e->top_xy_h, (size_t)MAP_TOP_POINTS * 2 * sizeof(float), hipHostMallocDefault) != hipSuccess)
    return;
  if (!e->top_ev && hipEventCreateWithFlags(&e->top_ev, hipEventDisableTiming) != hipSuccess) return;
  launch_collect_first(m.view, M, e->top_xy_d, e->s_aux);
  if (hipMemcpyAsync(e->top_xy_h, e->top_xy_d, (size_t)M * 2 * sizeof(float), hipMemcpyDeviceToHost, e->s_aux) !=
          hipSuccess ||
      hipEventRecord(e->top_ev, e->s_aux) != hipSuccess)
    return;
  const int dev = e->device;
  hipEvent_t ev = e->top_ev;
  const float *src = e->top_xy_h;
  DevMap *mp = &m;
  try {  // (no exception may leave the C ABI: without the helper the top is built on demand, ensure_map_top)
    m.top_thread = std::thread([dev, ev, src, mp, M] {
      (void)hipSetDevice(dev);
      if (hipEventSynchronize(ev) != hipSuccess) return;
      try {
        mp->top_xy.assign(src, src + (size_t)M * 2);
        insert_map_top(*mp, M);
      } catch (...) {
        mp->top_m = 0;
      }
    });
  } catch (...) {
  }
}

TrgStatus map_first_of_two(TrgEngine *e, DevMap &m, float qx, float qy, const TiePoint &A,
                           const TiePoint &B, int *first) {
  hipStream_t s = e->s_aux;  // (the main stream is busy with the next level's speculative expansion)
  TrgStatus st = ensure_map_top(e, m);
  if (st != TRG_OK) return st;
  MapTieWalk w{};
  w.key = ~0ull;
  w.lo[0] = w.lo[1] = -INFINITY;
  w.hi[0] = w.hi[1] = INFINITY;
  w.cur_perm = -1;
  w.axis = 0;
  w.qx = qx;
  w.qy = qy;
  w.aperm = A.perm;
  w.bperm = B.perm;
  w.ax = A.x;
  w.ay = A.y;
  w.bx = B.x;
  w.by = B.y;
  // the common path of A and B through the top of the tree, on the host (the same decisions as
  // region_step on the device, trg_kernels.hip)
  {
    int cur = 0;
    for (;;) {
      const float cx = m.top_xy[2 * (size_t)cur], cy = m.top_xy[2 * (size_t)cur + 1];
      const int axis = w.axis;
      const float split = axis ? cy : cx;
      const float q = axis ? qy : qx;
      const bool near_is_left = (q - split) <= 0;
      const float ca = axis ? A.y : A.x, cb = axis ? B.y : B.x;
      if (cur == A.perm || cur == B.perm) {
        const bool cur_is_a = cur == A.perm;
        const bool other_left = (cur_is_a ? cb : ca) < split;
        const bool other_first = other_left == near_is_left;
        *first = cur_is_a ? (other_first ? 1 : 0) : (other_first ? 0 : 1);
        return TRG_OK;
      }
      const bool a_left = ca < split, b_left = cb < split;
      if (a_left != b_left) {
        *first = (a_left == near_is_left) ? 0 : 1;
        return TRG_OK;
      }
      if (a_left)
        w.hi[axis] = split;
      else
        w.lo[axis] = split;
      w.cur_perm = cur;
      w.axis = axis ^ 1;
      w.steps++;
      const int child = a_left ? m.top_left[cur] : m.top_right[cur];
      if (child < 0) break;  // the subtree's root is a later point: the device goes on from this region
      cur = child;
    }
  }
  *e->mt_walk_h = w;
  HIPCHK(e, hipMemcpyAsync(e->mt_walk_d, e->mt_walk_h, sizeof(MapTieWalk), hipMemcpyHostToDevice, s));
  for (int batch = 0; batch < 64; ++batch) {
    // what is left of the region after the top of the tree holds ~N / 8192 points: one workgroup walks it
    // (a full-size region -- a map smaller than the top -- cannot get here)
    launch_map_tie_walk(m.view, e->mt_walk_d, 0, 64, s);
    HIPCHK(e, hipMemcpyAsync(e->mt_walk_h, e->mt_walk_d, sizeof(MapTieWalk), hipMemcpyDeviceToHost, s));
    HIPCHK(e, hipStreamSynchronize(s));
    if (e->mt_walk_h->done == 1) {
      *first = e->mt_walk_h->first;
      return TRG_OK;
    }
    if (e->mt_walk_h->done) break;
  }
  return e->fail(TRG_ERR_DEVICE, "nearest-point tie-break lost its candidates (internal error)");
}

// z of the map point kd_nearest returns for (qx, qy), ties decided as the reference's tree does
TrgStatus map_nn_exact(TrgEngine *e, DevMap &m, float qx, float qy, float *z, bool *found) {
  TrgStatus st = ensure_tie_scratch(e);
  if (st != TRG_OK) return st;
  hipStream_t s = e->s_aux;  // (the map is read-only here; the main stream may hold speculative work)
  launch_map_tied_set(m.view, qx, qy, e->prm.robot_size, e->mt_set_d, s);
  HIPCHK(e, hipMemcpyAsync(e->mt_set_h, e->mt_set_d, sizeof(MapTieSet), hipMemcpyDeviceToHost, s));
  HIPCHK(e, hipStreamSynchronize(s));
  const MapTieSet T = *e->mt_set_h;
  *found = T.count > 0;
  if (!*found) return TRG_OK;
  const int n = std::min(T.count, MAPTIE_SET_CAP);
  // lowest original index first, so that an unresolved case equals the hot kernels' provisional pick
  int order[MAPTIE_SET_CAP];
  for (int i = 0; i < n; ++i) order[i] = i;
  std::sort(order, order + n, [&](int a, int b) { return T.perm[a] < T.perm[b]; });
  *z = T.z[order[0]];
  if (T.count == 1) return TRG_OK;
  if (T.count > MAPTIE_SET_CAP) {
    e->stats.map_nn_unresolved++;
    return TRG_OK;
  }
  e->stats.map_nn_resolved++;
  bool same_z = true;
  for (int i = 1; i < n; ++i) same_z = same_z && T.z[order[i]] == T.z[order[0]];
  if (same_z) return TRG_OK;
  if (T.perm[order[0]] == 0) return TRG_OK;  // the root keeps an equal distance (kdtree.c:393-396)
  int w = order[0];
  for (int i = 1; i < n; ++i) {
    const int c = order[i];
    const TiePoint A{T.perm[w], T.x[w], T.y[w]}, B{T.perm[c], T.x[c], T.y[c]};
    int first = 0;
    st = map_first_of_two(e, m, qx, qy, A, B, &first);
    if (st != TRG_OK) return st;
    if (first == 1) w = c;
  }
  *z = T.z[w];
  return TRG_OK;
}

TrgStatus nearest_z_sync(TrgEngine *e, DevMap &m, const float *xy, size_t cnt, float *z,
                         int32_t *found) {
  if (!m.valid) return e->fail(TRG_ERR_NO_MAP, "nearest_z on an empty map");
  const size_t B = 1 << 20;
  for (size_t off = 0; off < cnt; off += B) {
    const size_t m_ = std::min(B, cnt - off);
    TrgStatus st = ensure_sync_scratch(e, m_);
    if (st != TRG_OK) return st;
    memcpy(e->sy_in.h, xy + 2 * off, m_ * 2 * sizeof(float));
    HIPCHK(e, hipMemcpyAsync(e->sy_in.d, e->sy_in.h, m_ * 2 * sizeof(float), hipMemcpyHostToDevice,
                             e->s_main));
    launch_probe_nearest_z(m.view, qparams(e), e->sy_in.d, (int)m_, e->sy_f0.d, e->sy_i0.d,
                           e->d_ctr, e->s_main);
    HIPCHK(e, hipMemcpyAsync(e->sy_f0.h, e->sy_f0.d, m_ * sizeof(float), hipMemcpyDeviceToHost,
                             e->s_main));
    HIPCHK(e, hipMemcpyAsync(e->sy_i0.h, e->sy_i0.d, m_ * sizeof(int), hipMemcpyDeviceToHost,
                             e->s_main));
    HIPCHK(e, hipStreamSynchronize(e->s_main));
    HIPCHK(e, hipGetLastError());
    memcpy(z + off, e->sy_f0.h, m_ * sizeof(float));
    // found == 2: several map points at the same fp32 distance; ask for the reference's choice
    std::vector<size_t> tied;
    for (size_t i = 0; i < m_; ++i) {
      if (e->sy_i0.h[i] == 2) tied.push_back(i);
      if (found) found[off + i] = e->sy_i0.h[i] ? 1 : 0;
    }
    for (size_t i : tied) {
      bool f = false;
      float zz = 0;
      st = map_nn_exact(e, m, xy[2 * (off + i)], xy[2 * (off + i) + 1], &zz, &f);
      if (st != TRG_OK) return st;
      if (f) z[off + i] = zz;
    }
  }
  e->stats.sync_batches++;
  return TRG_OK;
}

// The reference's slope gate (trg.cpp:269-274) evaluated with the host libm, used only for the
// sliver the device's exact rational test could not decide.
inline bool host_slope_gate(const TrgEngine *e, float z1, float z2, float dist) {
  float max_slope = atan2(e->prm.height_threshold, e->prm.robot_size);
  float slope = atan2(fabs(z1 - z2), dist);
  return slope > max_slope;
}
// final status code (0..4) of an edge evaluation after resolving an uncertain gate
inline int resolve_status(TrgEngine *e, int raw, float z1, float z2, float dist) {
  if (raw & EDGE_GATE_UNCERTAIN) {
    e->stats.gate_uncertain++;
    if (host_slope_gate(e, z1, z2, dist)) return EDGE_GATE;
  }
  return raw & EDGE_STATUS_MASK;
}

TrgStatus edges_sync(TrgEngine *e, DevMap &m, const float *p1, const float *p2, size_t cnt,
                     int32_t *status, int32_t *n_pts, float *weight, float *dist, bool resolve) {
  if (!m.valid) return e->fail(TRG_ERR_NO_MAP, "edge evaluation on an empty map");
  const size_t B = 1 << 18;
  for (size_t off = 0; off < cnt; off += B) {
    const size_t m_ = std::min(B, cnt - off);
    TrgStatus st = ensure_sync_scratch(e, m_);
    if (st != TRG_OK) return st;
    memcpy(e->sy_in.h, p1 + 3 * off, m_ * 3 * sizeof(float));
    memcpy(e->sy_in2.h, p2 + 3 * off, m_ * 3 * sizeof(float));
    HIPCHK(e, hipMemcpyAsync(e->sy_in.d, e->sy_in.h, m_ * 3 * sizeof(float), hipMemcpyHostToDevice,
                             e->s_main));
    HIPCHK(e, hipMemcpyAsync(e->sy_in2.d, e->sy_in2.h, m_ * 3 * sizeof(float),
                             hipMemcpyHostToDevice, e->s_main));
    launch_edges(m.view, qparams(e), e->sy_in.d, e->sy_in2.d, (int)m_, e->sy_mid, e->sy_i0.d,
                 e->sy_i1.d, e->sy_f0.d, e->sy_f1.d, e->d_ctr, e->s_main);
    HIPCHK(e, hipMemcpyAsync(e->sy_i0.h, e->sy_i0.d, m_ * sizeof(int), hipMemcpyDeviceToHost,
                             e->s_main));
    HIPCHK(e, hipMemcpyAsync(e->sy_i1.h, e->sy_i1.d, m_ * sizeof(int), hipMemcpyDeviceToHost,
                             e->s_main));
    HIPCHK(e, hipMemcpyAsync(e->sy_f0.h, e->sy_f0.d, m_ * sizeof(float), hipMemcpyDeviceToHost,
                             e->s_main));
    HIPCHK(e, hipMemcpyAsync(e->sy_f1.h, e->sy_f1.d, m_ * sizeof(float), hipMemcpyDeviceToHost,
                             e->s_main));
    HIPCHK(e, hipStreamSynchronize(e->s_main));
    HIPCHK(e, hipGetLastError());
    for (size_t i = 0; i < m_; ++i) {
      int raw = e->sy_i0.h[i];
      int stt = raw;
      if (resolve) {
        stt = resolve_status(e, raw, p1[3 * (off + i) + 2], p2[3 * (off + i) + 2], e->sy_f1.h[i]);
      }
      if (status) status[off + i] = stt;
      if (n_pts) n_pts[off + i] = e->sy_i1.h[i];
      if (weight) weight[off + i] = (stt == EDGE_OK) ? e->sy_f0.h[i] : 0.0f;
      if (dist) dist[off + i] = e->sy_f1.h[i];
    }
    e->stats.edge_evals_gpu += m_;
  }
  e->stats.sync_batches++;
  return TRG_OK;
}

