// Part of trg_engine.cpp (included inside its anonymous namespace): the map index build (TRG::setGlobalMap /
// setLocalMap, trg.cpp:179-209 -> cell-sorted arrays) and the staged upload of a host cloud.
// ---- map index build ---------------------------------------------------------------------------
void start_map_top(TrgEngine *e, DevMap &m);

TrgStatus build_map(TrgEngine *e, DevMap &m, const float *d_xyz, size_t n, size_t stride) {
  m.top_wait();  // (a helper of the previous build still reads the arrays that are replaced below)
  auto t_host = Clock::now();
  m.valid = false;
  if (n == 0) {
    m.n = 0;
    return TRG_OK;
  }
  if (n > (size_t)0x7FFFFFF0) return e->fail(TRG_ERR_CAPACITY, "more than 2^31 map points");
  hipStream_t s = e->s_main;
  hipEvent_t ev0, ev1;
  HIPCHK(e, hipEventCreate(&ev0));
  HIPCHK(e, hipEventCreate(&ev1));
  HIPCHK(e, hipEventRecord(ev0, s));
  launch_init_bounds(e->d_bounds, s);
  launch_bounds(d_xyz, n, stride, e->d_bounds, s);
  unsigned hb[4];
  HIPCHK(e, hipMemcpyAsync(hb, e->d_bounds, sizeof(hb), hipMemcpyDeviceToHost, s));
  HIPCHK(e, hipStreamSynchronize(s));
  const float x0 = key_to_float(hb[0]), y0 = key_to_float(hb[1]);
  const float x1 = key_to_float(hb[2]), y1 = key_to_float(hb[3]);
  if (!(x1 >= x0) || !(y1 >= y0) || !std::isfinite(x0) || !std::isfinite(x1) ||
      !std::isfinite(y0) || !std::isfinite(y1)) {
    return e->fail(TRG_ERR_INVALID_ARG, "map has non-finite coordinates");
  }
  // cell size = robot_size: a collision disc touches a 3x3 block, an edge ellipse <= 7x7
  float g = e->prm.robot_size;
  if (!(g > 0)) g = 0.3f;
  const double max_cells = 64.0 * 1024 * 1024;
  while (((double)(x1 - x0) / g + 2) * ((double)(y1 - y0) / g + 2) > max_cells) g *= 2.0f;
  const float inv_g = 1.0f / g;
  const int W = (int)floorf((x1 - x0) * inv_g) + 1;
  const int H = (int)floorf((y1 - y0) * inv_g) + 1;
  const size_t ncell = (size_t)W * H;

  if (m.cap_pts < n) {
    if (m.x) (void)hipFree(m.x);
    if (m.y) (void)hipFree(m.y);
    if (m.z) (void)hipFree(m.z);
    if (m.perm) (void)hipFree(m.perm);
    if (m.pt) (void)hipFree(m.pt);
    m.x = m.y = m.z = nullptr;
    m.perm = nullptr;
    m.pt = nullptr;
    HIPCHK(e, hipMalloc((void **)&m.pt, n * sizeof(float4)));
    HIPCHK(e, hipMalloc((void **)&m.x, n * sizeof(float)));
    HIPCHK(e, hipMalloc((void **)&m.y, n * sizeof(float)));
    HIPCHK(e, hipMalloc((void **)&m.z, n * sizeof(float)));
    HIPCHK(e, hipMalloc((void **)&m.perm, n * sizeof(int)));
    m.cap_pts = n;
  }
  if (m.cap_cells < ncell + 1) {
    if (m.cell_start) (void)hipFree(m.cell_start);
    m.cell_start = nullptr;
    HIPCHK(e, hipMalloc((void **)&m.cell_start, (ncell + 1) * sizeof(int)));
    m.cap_cells = ncell + 1;
  }
  // scratch of the build, kept with the engine (allocating and freeing 240 MB per build costs as much as
  // a kernel of it)
  IndexScratch &sc = e->idx_scratch;
  if (sc.cap_pts < n) {
    if (sc.aos) (void)hipFree(sc.aos);
    sc.aos = nullptr;
    sc.cap_pts = 0;
    HIPCHK(e, hipMalloc((void **)&sc.aos, n * 16));
    sc.cap_pts = n;
  }
  int bin_shift = 0, nbins = 0, nwg = 0;
  if (!getenv("TRG_INDEX_DIRECT") && index_bins_plan(n, ncell, &bin_shift, &nbins, &nwg)) {
    // through bins of ~one cell row (trg_kernels.hip): no global atomics, no random line per point
    const size_t nb = (size_t)nbins * nwg;
    if (sc.cap_bins < nb) {
      if (sc.hist) (void)hipFree(sc.hist);
      if (sc.base) (void)hipFree(sc.base);
      if (sc.bin_tmp) (void)hipFree(sc.bin_tmp);
      sc.hist = sc.base = sc.bin_tmp = nullptr;
      sc.cap_bins = 0;
      HIPCHK(e, hipMalloc((void **)&sc.hist, (nb + 1) * sizeof(int)));
      HIPCHK(e, hipMalloc((void **)&sc.base, (nb + 1) * sizeof(int)));
      HIPCHK(e, hipMalloc((void **)&sc.bin_tmp, (nb / 2048 + 4) * sizeof(int)));
      sc.cap_bins = nb;
    }
    // (the map's own record array is the first scratch: it is rewritten by the last kernel)
    launch_index_bins(d_xyz, n, stride, x0, y0, inv_g, W, H, (int)ncell, bin_shift, nbins, nwg, sc.hist, sc.base,
                      sc.bin_tmp, m.pt, (float4 *)sc.aos, m.cell_start, m.x, m.y, m.z, m.perm, m.pt, s);
  } else {
    if (sc.cap_direct < n) {
      if (sc.cell_of) (void)hipFree(sc.cell_of);
      if (sc.rank) (void)hipFree(sc.rank);
      sc.cell_of = sc.rank = nullptr;
      sc.cap_direct = 0;
      HIPCHK(e, hipMalloc((void **)&sc.cell_of, n * sizeof(int)));
      HIPCHK(e, hipMalloc((void **)&sc.rank, n * sizeof(int)));
      sc.cap_direct = n;
    }
    if (sc.cap_cells < ncell) {
      if (sc.counts) (void)hipFree(sc.counts);
      if (sc.tmp) (void)hipFree(sc.tmp);
      sc.counts = sc.tmp = nullptr;
      sc.cap_cells = 0;
      HIPCHK(e, hipMalloc((void **)&sc.counts, ncell * sizeof(int)));
      HIPCHK(e, hipMalloc((void **)&sc.tmp, (ncell / 2048 + 4) * sizeof(int)));
      sc.cap_cells = ncell;
    }
    int *d_cell_of = sc.cell_of, *d_rank = sc.rank, *d_counts = sc.counts, *d_tmp = sc.tmp;
    HIPCHK(e, hipMemsetAsync(d_counts, 0, ncell * sizeof(int), s));
    launch_cell_count(d_xyz, n, stride, x0, y0, inv_g, W, H, d_cell_of, d_rank, d_counts, s);
    launch_exclusive_scan(d_counts, m.cell_start, (int)ncell, d_tmp, s);
    launch_scatter_sort_aos(d_xyz, n, stride, d_cell_of, d_rank, (int)ncell, m.cell_start, sc.aos, m.x, m.y, m.z,
                            m.perm, m.pt, s);
  }
  HIPCHK(e, hipEventRecord(ev1, s));
  HIPCHK(e, hipStreamSynchronize(s));
  HIPCHK(e, hipGetLastError());
  float ms = 0;
  (void)hipEventElapsedTime(&ms, ev0, ev1);
  (void)hipEventDestroy(ev0);
  (void)hipEventDestroy(ev1);

  m.n = n;
  m.g = g;
  m.bounds[0] = x0;
  m.bounds[1] = y0;
  m.bounds[2] = x1;
  m.bounds[3] = y1;
  m.view.x = m.x;
  m.view.y = m.y;
  m.view.z = m.z;
  m.view.pt = m.pt;
  m.view.perm = m.perm;
  m.view.cell_start = m.cell_start;
  m.view.x0 = x0;
  m.view.y0 = y0;
  m.view.inv_g = inv_g;
  m.view.W = W;
  m.view.H = H;
  m.view.n = (int)n;
  m.valid = true;
  m.top_wait();
  m.top_m = 0;  // (the top of the insertion tree: on demand, for the global map beside the build)
  if (&m == &e->gmap) {
    e->stats.map_points = n;
    e->stats.ms_index_build = ms;
    // read xyz once, write the cell-sorted SoA once, cell ids once
    e->stats.bytes_index_build = (uint64_t)(12 + 12 + 4) * n;
    e->stats.ms_set_map_total = ms_since(t_host);
    start_map_top(e, m);
  }
  return TRG_OK;
}

// Host cloud -> HBM.  TRG::setGlobalMap / setLocalMap get a cloud in ordinary (pageable) host memory
// (trg.cpp:179-193, 195-209); a plain hipMemcpy from there runs at a third of the link rate (the runtime
// stages it through one pinned buffer on one thread: ~25 ms for the 120 MB of C3).  Here UP_THREADS host
// threads copy interleaved chunks into pinned staging slots of their own and send every chunk on with
// hipMemcpyAsync on a stream of their own, so the CPU copies and the DMA transfers overlap; a source that
// is already pinned (hipHostMalloc / hipHostRegister / a pinned torch tensor) goes out in one async copy.
constexpr int UP_THREADS = 4, UP_SLOTS = 2;
constexpr size_t UP_CHUNK = (size_t)8 << 20;
struct Uploader {
  char *pinned = nullptr;  // UP_THREADS * UP_SLOTS chunks
  hipStream_t st[UP_THREADS] = {};
  hipEvent_t ev[UP_THREADS][UP_SLOTS] = {};
  float *d_in = nullptr;   // device staging of the raw cloud (kept across calls)
  size_t d_cap = 0;
  void release() {
    if (pinned) (void)hipHostFree(pinned);
    for (auto &s : st)
      if (s) (void)hipStreamDestroy(s);
    for (auto &row : ev)
      for (auto &x : row)
        if (x) (void)hipEventDestroy(x);
    if (d_in) (void)hipFree(d_in);
    *this = Uploader();
  }
};

TrgStatus staged_upload(TrgEngine *e, void *d_dst, const void *src, size_t bytes) {
  Uploader &u = *e->uploader;
  hipPointerAttribute_t attr;
  const bool pinned_src = hipPointerGetAttributes(&attr, src) == hipSuccess && attr.type == hipMemoryTypeHost;
  (void)hipGetLastError();  // (an ordinary malloc pointer makes the query fail: not an error)
  if (pinned_src || bytes < UP_CHUNK) {
    HIPCHK(e, hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, e->s_main));
    HIPCHK(e, hipStreamSynchronize(e->s_main));
    return TRG_OK;
  }
  if (!u.pinned) {
    HIPCHK(e, hipHostMalloc((void **)&u.pinned, UP_CHUNK * UP_THREADS * UP_SLOTS, hipHostMallocDefault));
    for (int t = 0; t < UP_THREADS; ++t) {
      HIPCHK(e, hipStreamCreateWithFlags(&u.st[t], hipStreamNonBlocking));
      for (int k = 0; k < UP_SLOTS; ++k) HIPCHK(e, hipEventCreateWithFlags(&u.ev[t][k], hipEventDisableTiming));
    }
  }
  const size_t nchunk = (bytes + UP_CHUNK - 1) / UP_CHUNK;
  std::atomic<int> bad{0};
  auto work = [&](int t) {
    if (hipSetDevice(e->device) != hipSuccess) {
      bad = 1;
      return;
    }
    int use = 0;
    for (size_t c = (size_t)t; c < nchunk; c += UP_THREADS, ++use) {
      const int k = use % UP_SLOTS;
      char *slot = u.pinned + ((size_t)t * UP_SLOTS + k) * UP_CHUNK;
      if (use >= UP_SLOTS && hipEventSynchronize(u.ev[t][k]) != hipSuccess) bad = 1;  // the slot's last transfer
      const size_t off = c * UP_CHUNK, len = std::min(UP_CHUNK, bytes - off);
      memcpy(slot, (const char *)src + off, len);
      if (hipMemcpyAsync((char *)d_dst + off, slot, len, hipMemcpyHostToDevice, u.st[t]) != hipSuccess) bad = 1;
      if (hipEventRecord(u.ev[t][k], u.st[t]) != hipSuccess) bad = 1;
    }
    if (hipStreamSynchronize(u.st[t]) != hipSuccess) bad = 1;
  };
  std::vector<std::thread> thr;
  for (int t = 1; t < UP_THREADS; ++t) thr.emplace_back(work, t);
  work(0);
  for (auto &th : thr) th.join();
  if (bad) return e->fail(TRG_ERR_DEVICE, "staged upload of the cloud failed");
  return TRG_OK;
}

TrgStatus upload_and_build(TrgEngine *e, DevMap &m, const float *xyz, size_t n, size_t stride) {
  if (n == 0) {
    m.n = 0;
    m.valid = false;
    return TRG_OK;
  }
  Uploader &u = *e->uploader;
  const size_t floats = n * stride;
  if (u.d_cap < floats) {
    if (u.d_in) (void)hipFree(u.d_in);
    u.d_in = nullptr;
    u.d_cap = 0;
    HIPCHK(e, hipMalloc((void **)&u.d_in, floats * sizeof(float)));
    u.d_cap = floats;
  }
  auto t0 = Clock::now();
  TrgStatus st = staged_upload(e, u.d_in, xyz, floats * sizeof(float));
  e->stats.ms_upload = ms_since(t0);
  if (st != TRG_OK) return st;
  return build_map(e, m, u.d_in, n, stride);
}

