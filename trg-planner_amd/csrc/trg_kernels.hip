// trg_kernels.hip -- hand-written gfx950 (CDNA4) kernels of the TRG construction hot path.
//
// What each kernel reproduces (reference = /root/reference/cpp/trg_planner/core/trg_planner):
//   index build   <- TRG::setGlobalMap's kd_insert2 loop            src/graph/trg.cpp:185-188
//   disc_query    <- kd_nearest_range2 + TRG::isCollision            src/kdtree/kdtree.c:270-301, trg.cpp:746-778
//                    (+ the kd_nearest2 elevation lookup, trg.cpp:244-247, fused when asked)
//   edge_gather   <- the position-only part of TRG::wireEdge          trg.cpp:269-363
//   (+ k_edge_finish: covariance -> JacobiSVD -> weight, trg.cpp:339-363)
//   sample_nodes  <- the rejection sampling loop of TRG::expandGraph  trg.cpp:384-403
//
// Execution model: one 64-lane wavefront per query.  A query's candidate points are the points
// of a few cell rows of the cell-sorted SoA map; each row segment is ONE contiguous range, so the
// 64 lanes issue coalesced loads of x[], y[], z[].  Hits are compacted with wave ballot +
// popcount prefix into a per-wave LDS tile, the median is a rank selection over that tile.
// No MFMA: nothing here is a dense contraction (the only "matrix" is a 3x3 covariance).
//
// Floating point: compiled with -ffp-contract=off; every fp32 decision (inclusive radius test,
// ellipse test, segment walk accumulation, division/sqrt) is evaluated with the same operations
// in the same order as the reference's x86-64 (no FMA) build, so decisions are bit-identical.
#include "trg_kernels.h"

#include <float.h>
#include <limits.h>

namespace trg {

namespace {

constexpr int WAVE = 64;
constexpr int HCAP = 1024;  // hits kept in the per-wave LDS tile; larger discs use the fallback

__device__ __forceinline__ int lane_id() { return threadIdx.x & (WAVE - 1); }

__device__ __forceinline__ unsigned long long lanemask_lt() {
  return (1ull << lane_id()) - 1ull;
}

// ballot of a bool without the int round trip of __ballot()
__device__ __forceinline__ unsigned long long ballot(bool pred) {
  return __builtin_amdgcn_ballot_w64(pred);
}

__device__ __forceinline__ void wave_lds_sync() {
  // LDS traffic of one wave is in order; this only stops the compiler from moving accesses
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}

__device__ __forceinline__ int cell_coord(float v, float v0, float inv_g, int ncell) {
  float t = floorf((v - v0) * inv_g);
  // clamp in float first: the int conversion of a huge value is undefined
  t = fminf(fmaxf(t, 0.0f), (float)(ncell - 1));
  return (int)t;
}

__device__ __forceinline__ unsigned float_key(float f) {
  unsigned b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key_float(unsigned k) {
  unsigned b = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
  return __uint_as_float(b);
}

struct CellRange {
  int cx0, cx1, cy0, cy1;
};
__device__ __forceinline__ CellRange cells_for(const MapView &m, float qx, float qy, float r) {
  // conservative: the exact fp32 test decides membership, this only bounds the candidates
  float rp = r * 1.001f + 1e-6f;
  CellRange c;
  c.cx0 = cell_coord(qx - rp, m.x0, m.inv_g, m.W);
  c.cx1 = cell_coord(qx + rp, m.x0, m.inv_g, m.W);
  c.cy0 = cell_coord(qy - rp, m.y0, m.inv_g, m.H);
  c.cy1 = cell_coord(qy + rp, m.y0, m.inv_g, m.H);
  return c;
}

struct Disc {
  int n;        // points with d2 <= r2
  int cnt;      // of those, |z - z_med| > height_threshold
  float nn_z;   // z of the nearest point (only when NN)
  int nn_tie;   // another point had the same fp32 d2 as the nearest
  float nn_d2;  // that distance
};

// Large-disc fallback: k-th smallest z by radix selection over the candidates, re-read from the
// (L2-resident) map; no LDS tile needed.  Correct for any density, only slower.
__device__ float select_kth_global(const MapView &m, const CellRange &c, float qx, float qy,
                                   float r2, int k) {
  unsigned prefix = 0;
  for (int bit = 31; bit >= 0; --bit) {
    unsigned cand = prefix | (1u << bit);
    int below = 0;
    for (int cy = c.cy0; cy <= c.cy1; ++cy) {
      int s = m.cell_start[cy * m.W + c.cx0];
      int e = m.cell_start[cy * m.W + c.cx1 + 1];
      for (int base = s; base < e; base += WAVE) {
        int i = base + lane_id();
        if (i < e) {
          float dx = m.x[i] - qx, dy = m.y[i] - qy;
          float d2 = dx * dx + dy * dy;
          if (d2 <= r2 && float_key(m.z[i]) < cand) below++;
        }
      }
    }
    below = wave_sum(below);
    if (below <= k) prefix = cand;
  }
  return key_float(prefix);
}

// One wave: all map points within r of (qx, qy) in 2-D, inclusive fp32 test with the reference's
// operation order (kdtree.c:277-281), then the isCollision statistics (trg.cpp:763-772).
template <bool NN>
__device__ Disc disc_query(const MapView &m, float qx, float qy, float r, float h, float *zbuf,
                           DeviceCounters *ctr, int cap = HCAP) {
  const int lane = lane_id();
  const float r2 = r * r;
  const CellRange c = cells_for(m, qx, qy, r);
  int n = 0;
  float best_d2 = FLT_MAX, best_z = 0.0f;
  int best_perm = INT_MAX;
  bool lane_tie = false;  // this lane saw two points at its own best distance
  for (int cy = c.cy0; cy <= c.cy1; ++cy) {
    const int s = m.cell_start[cy * m.W + c.cx0];
    const int e = m.cell_start[cy * m.W + c.cx1 + 1];
    for (int base = s; base < e; base += WAVE) {
      const int i = base + lane;
      bool hit = false;
      float z = 0.0f, d2 = 0.0f;
      if (i < e) {
        const float dx = m.x[i] - qx;
        const float dy = m.y[i] - qy;
        z = m.z[i];
        d2 = dx * dx + dy * dy;
        hit = d2 <= r2;
      }
      const unsigned long long mask = __ballot(hit);
      if (hit) {
        const int pos = n + __popcll(mask & lanemask_lt());
        if (pos < cap) zbuf[pos] = z;
        if (NN) {
          const int pm = m.perm[i];
          if (d2 == best_d2) lane_tie = true;
          if (d2 < best_d2) lane_tie = false;
          if (d2 < best_d2 || (d2 == best_d2 && pm < best_perm)) {
            best_d2 = d2;
            best_z = z;
            best_perm = pm;
          }
        }
      }
      n += __popcll(mask);
    }
  }
  Disc out;
  out.n = n;
  out.cnt = 0;
  out.nn_z = 0.0f;
  out.nn_tie = 0;
  out.nn_d2 = 0.0f;
  if (n == 0) return out;

  if (NN) {
    // wave arg-min over (d2, perm); afterwards every lane holds the winner
    float wd = best_d2, wz = best_z;
    int wp = best_perm;
#pragma unroll
    for (int msk = 32; msk >= 1; msk >>= 1) {
      const float od = __shfl_xor(wd, msk);
      const float oz = __shfl_xor(wz, msk);
      const int op = __shfl_xor(wp, msk);
      if (od < wd || (od == wd && op < wp)) {
        wd = od;
        wz = oz;
        wp = op;
      }
    }
    out.nn_z = wz;
    out.nn_d2 = wd;
    // a lane whose own best ties the winner with a different point
    out.nn_tie =
        __ballot(best_perm != INT_MAX && best_d2 == wd && (best_perm != wp || lane_tie)) != 0ull;
  }

  float zmed;
  const int k = n / 2;  // pts[pts.size() / 2] after the ascending sort (trg.cpp:764)
  if (n <= cap) {
    wave_lds_sync();
    float mine = 0.0f;
    bool found = false;
    for (int i = lane; i < n; i += WAVE) {
      const float zi = zbuf[i];
      int rank = 0;
      for (int j = 0; j < n; ++j) {
        const float zj = zbuf[j];
        rank += (zj < zi) || (zj == zi && j < i);
      }
      if (rank == k) {
        mine = zi;
        found = true;
      }
    }
    const unsigned long long who = __ballot(found);
    zmed = __shfl(mine, __ffsll((long long)who) - 1);
    int cnt = 0;
    for (int i = lane; i < n; i += WAVE) cnt += fabsf(zbuf[i] - zmed) > h;
    out.cnt = wave_sum(cnt);
    wave_lds_sync();  // tile is reused by the next query of this wave
  } else {
    if (ctr && lane == 0) atomicAdd(&ctr[blockIdx.x % COUNTER_SHARDS].overflow, 1ull);
    zmed = select_kth_global(m, c, qx, qy, r2, k);
    int cnt = 0;
    for (int cy = c.cy0; cy <= c.cy1; ++cy) {
      const int s = m.cell_start[cy * m.W + c.cx0];
      const int e = m.cell_start[cy * m.W + c.cx1 + 1];
      for (int base = s; base < e; base += WAVE) {
        const int i = base + lane;
        if (i < e) {
          const float dx = m.x[i] - qx, dy = m.y[i] - qy;
          const float d2 = dx * dx + dy * dy;
          if (d2 <= r2 && fabsf(m.z[i] - zmed) > h) cnt++;
        }
      }
    }
    out.cnt = wave_sum(cnt);
  }
  return out;
}

__device__ __forceinline__ bool disc_collides(const Disc &d, float threshold) {
  if (d.n == 0) return true;                       // trg.cpp:749-752
  const float ratio = (float)d.cnt / (float)d.n;   // trg.cpp:773
  return ratio > threshold;
}

// Nearest map point in 2-D for an arbitrary position: grow a square window of cells until the
// best candidate is provably the nearest (everything outside the window is farther than R).
__device__ bool nearest_point(const MapView &m, float qx, float qy, float r0, float &z_out,
                              int &tie_out) {
  const int lane = lane_id();
  float R = r0;
  for (int iter = 0; iter < 40; ++iter) {
    const CellRange c = cells_for(m, qx, qy, R);
    float best_d2 = FLT_MAX, best_z = 0.0f;
    int best_perm = INT_MAX;
    bool lane_tie = false;
    for (int cy = c.cy0; cy <= c.cy1; ++cy) {
      const int s = m.cell_start[cy * m.W + c.cx0];
      const int e = m.cell_start[cy * m.W + c.cx1 + 1];
      for (int i = s + lane; i < e; i += WAVE) {
        const float dx = m.x[i] - qx, dy = m.y[i] - qy;
        const float d2 = dx * dx + dy * dy;
        const int pm = m.perm[i];
        if (d2 == best_d2) lane_tie = true;
        if (d2 < best_d2) lane_tie = false;
        if (d2 < best_d2 || (d2 == best_d2 && pm < best_perm)) {
          best_d2 = d2;
          best_z = m.z[i];
          best_perm = pm;
        }
      }
    }
    float wd = best_d2, wz = best_z;
    int wp = best_perm;
#pragma unroll
    for (int msk = 32; msk >= 1; msk >>= 1) {
      const float od = __shfl_xor(wd, msk);
      const float oz = __shfl_xor(wz, msk);
      const int op = __shfl_xor(wp, msk);
      if (od < wd || (od == wd && op < wp)) {
        wd = od;
        wz = oz;
        wp = op;
      }
    }
    const bool whole = c.cx0 == 0 && c.cy0 == 0 && c.cx1 == m.W - 1 && c.cy1 == m.H - 1;
    if (wp != INT_MAX && (wd <= R * R || whole)) {
      z_out = wz;
      tie_out =
          __ballot(best_perm != INT_MAX && best_d2 == wd && (best_perm != wp || lane_tie)) != 0ull;
      return true;
    }
    if (whole) return false;
    R *= 2.0f;
  }
  return false;
}

// ---- 3x3 SVD (U only), fp32: the published two-sided Jacobi of Eigen 3.4's JacobiSVD, which is
// what the reference calls at trg.cpp:339 (Eigen/src/SVD/JacobiSVD.h, Eigen/src/Jacobi/Jacobi.h).
struct Rot {
  float c, s;
};
__device__ __forceinline__ Rot rot_mul(Rot a, Rot b) {
  Rot r;
  r.c = a.c * b.c - a.s * b.s;
  r.s = a.c * b.s + a.s * b.c;
  return r;
}
__device__ __forceinline__ Rot rot_t(Rot a) {
  Rot r;
  r.c = a.c;
  r.s = -a.s;
  return r;
}
__device__ __forceinline__ void rows_rotate(float M[3][3], int p, int q, Rot j) {
  if (j.c == 1.0f && j.s == 0.0f) return;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float xi = M[p][i], yi = M[q][i];
    M[p][i] = j.c * xi + j.s * yi;
    M[q][i] = -j.s * xi + j.c * yi;
  }
}
__device__ __forceinline__ void cols_rotate(float M[3][3], int p, int q, Rot j) {
  const Rot t = rot_t(j);
  if (t.c == 1.0f && t.s == 0.0f) return;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float xi = M[i][p], yi = M[i][q];
    M[i][p] = t.c * xi + t.s * yi;
    M[i][q] = -t.s * xi + t.c * yi;
  }
}
__device__ __forceinline__ void make_jacobi(float x, float y, float z, Rot &r) {
  const float deno = 2.0f * fabsf(y);
  if (deno < FLT_MIN) {
    r.c = 1.0f;
    r.s = 0.0f;
    return;
  }
  const float tau = (x - z) / deno;
  const float w = sqrtf(tau * tau + 1.0f);
  float t;
  if (tau > 0.0f) {
    t = 1.0f / (tau + w);
  } else {
    t = 1.0f / (tau - w);
  }
  const float sign_t = t > 0.0f ? 1.0f : -1.0f;
  const float n = 1.0f / sqrtf(t * t + 1.0f);
  r.s = -sign_t * (y / fabsf(y)) * fabsf(t) * n;
  r.c = n;
}
__device__ __forceinline__ void jacobi_2x2(float W[3][3], int p, int q, Rot &jl, Rot &jr) {
  float m00 = W[p][p], m01 = W[p][q], m10 = W[q][p], m11 = W[q][q];
  Rot rot1;
  const float t = m00 + m11;
  const float d = m10 - m01;
  if (fabsf(d) < FLT_MIN) {
    rot1.s = 0.0f;
    rot1.c = 1.0f;
  } else {
    const float u = t / d;
    const float tmp = sqrtf(1.0f + u * u);
    rot1.s = 1.0f / tmp;
    rot1.c = u / tmp;
  }
  if (!(rot1.c == 1.0f && rot1.s == 0.0f)) {
    const float a0 = m00, b0 = m10, a1 = m01, b1 = m11;
    m00 = rot1.c * a0 + rot1.s * b0;
    m10 = -rot1.s * a0 + rot1.c * b0;
    m01 = rot1.c * a1 + rot1.s * b1;
    m11 = -rot1.s * a1 + rot1.c * b1;
  }
  make_jacobi(m00, m01, m11, jr);
  jl = rot_mul(rot1, rot_t(jr));
}
__device__ void svd_u3(const float A[3][3], float U[3][3]) {
  float W[3][3];
  float scale = 0.0f;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) scale = fmaxf(scale, fabsf(A[r][c]));
  if (scale == 0.0f) scale = 1.0f;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      W[r][c] = A[r][c] / scale;
      U[r][c] = (r == c) ? 1.0f : 0.0f;
    }
  const float precision = 2.0f * FLT_EPSILON;
  float max_diag = fmaxf(fabsf(W[0][0]), fmaxf(fabsf(W[1][1]), fabsf(W[2][2])));
  bool finished = false;
  for (int sweep = 0; sweep < 1000 && !finished; ++sweep) {
    finished = true;
#pragma unroll
    for (int p = 1; p < 3; ++p) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        if (q >= p) continue;
        const float threshold = fmaxf(FLT_MIN, precision * max_diag);
        if (fabsf(W[p][q]) > threshold || fabsf(W[q][p]) > threshold) {
          finished = false;
          Rot jl, jr;
          jacobi_2x2(W, p, q, jl, jr);
          rows_rotate(W, p, q, jl);
          cols_rotate(U, p, q, rot_t(jl));
          cols_rotate(W, p, q, jr);
          max_diag = fmaxf(max_diag, fmaxf(fabsf(W[p][p]), fabsf(W[q][q])));
        }
      }
    }
  }
  float sv[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float a = fabsf(W[i][i]);
    sv[i] = a;
    if (a != 0.0f) {
      const float sgn = W[i][i] / a;
#pragma unroll
      for (int r = 0; r < 3; ++r) U[r][i] *= sgn;
    }
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) sv[i] *= scale;
  // sort by decreasing singular value (selection, first maximum wins)
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    int pos = i;
    float mx = sv[i];
#pragma unroll
    for (int k2 = 0; k2 < 3; ++k2) {
      if (k2 > i && sv[k2] > mx) {
        mx = sv[k2];
        pos = k2;
      }
    }
    if (mx == 0.0f) break;
    if (pos != i) {
      const float ts = sv[i];
      sv[i] = sv[pos];
      sv[pos] = ts;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const float tu = U[r][i];
        U[r][i] = U[r][pos];
        U[r][pos] = tu;
      }
    }
  }
}

// covariance -> risk weight: JacobiSVD(cov).matrixU().normalized(), |z components| of the two
// leading singular vectors, 0.8/0.2 blend, < 0.1 clamp (trg.cpp:339-363)
__device__ float risk_weight(const float cov[3][3], int &clamped) {
  float U[3][3];
  svd_u3(cov, U);
  float fro = 0.0f;
#pragma unroll
  for (int cc = 0; cc < 3; ++cc)
#pragma unroll
    for (int r = 0; r < 3; ++r) fro += U[r][cc] * U[r][cc];
  float e20 = U[2][0], e21 = U[2][1];
  if (fro > 0.0f) {
    const float nrm = sqrtf(fro);
    e20 = e20 / nrm;
    e21 = e21 / nrm;
  }
  float hor = e20, ver = e21;  // col(k).dot(-gravity), trg.cpp:347-354
  if (hor < 0.0f) hor = -e20;
  if (ver < 0.0f) ver = -e21;
  const float ratio = 0.8f;
  float w = ratio * hor + (1 - ratio) * ver;
  clamped = 0;
  if ((double)w < 0.1) {
    w = 0.0f;
    clamped = EDGE_CLAMPED;
  }
  return w;
}

// ---- edge evaluation -----------------------------------------------------------------------------
// Phase 1 (one wave per edge) produces the "mid" record: status, point count, dist and the 3x3
// covariance; phase 2 (k_edge_finish, one THREAD per edge) runs the SVD.  Splitting them keeps the
// serial ~500-flop Jacobi chain off the 64-lane gather waves.
// dwords per mid record: status, n_pts, dist, hits, then the nine fp64 moments of the kept ellipse
// points (s_x s_y s_z s_xx s_xy s_xz s_yy s_yz s_zz; z relative to node 1), 2 dwords of padding.
// The covariance is formed by the thread-per-edge finish kernels (mid_covariance).
constexpr int MID_STRIDE = 24;
constexpr int MID_HITS = 3;
constexpr int MID_MOMENTS = 4;

struct EdgeGeom {
  float dist, dirx, diry, cx, cy;
  int uncertain;
  bool gated;
};

__device__ __forceinline__ EdgeGeom edge_geometry(const QueryParams &p, float x1, float y1, float z1,
                                                  float x2, float y2, float z2) {
  EdgeGeom g;
  const float ex = x1 - x2, ey = y1 - y2;
  g.dist = sqrtf(ex * ex + ey * ey);  // (node1.head(2) - node2.head(2)).norm()
  // slope gate, trg.cpp:269-274: atan2f(|dz|, dist) > atan2f(h, r).  atan2 is monotone in the
  // ratio, so the exact rational comparison |dz| * r  vs  h * dist decides it whenever the two
  // sides differ by more than 1e-4 relative (fp32 atan2f error is ~1e-7); the sliver in between
  // is flagged and the host applies the reference's own libm comparison.
  g.uncertain = 0;
  g.gated = false;
  const double lhs = (double)fabsf(z1 - z2) * (double)p.robot_size;
  const double rhs = (double)p.height_threshold * (double)g.dist;
  const double margin = (double)p.gate_margin;
  if (lhs > rhs * (1.0 + margin)) {
    g.gated = true;
  } else if (!(lhs < rhs * (1.0 - margin))) {
    g.uncertain = EDGE_GATE_UNCERTAIN;
  }
  // dir = (node2 - node1).normalized(); center = node1 + 0.5 * dist * dir  (trg.cpp:277-278)
  const float dx = x2 - x1, dy = y2 - y1;
  const float sq = dx * dx + dy * dy;
  g.dirx = dx;
  g.diry = dy;
  if (sq > 0.0f) {
    const float nrm = sqrtf(sq);
    g.dirx = dx / nrm;
    g.diry = dy / nrm;
  }
  const float half = 0.5f * g.dist;
  g.cx = x1 + half * g.dirx;
  g.cy = y1 + half * g.diry;
  return g;
}

struct Moments {
  double s_x = 0, s_y = 0, s_z = 0, s_xx = 0, s_xy = 0, s_xz = 0, s_yy = 0, s_yz = 0, s_zz = 0;
  int kept = 0, in_range = 0;
};

struct EllipseParams {
  float cx, cy, a2, bb, aabb, r00, r01, r10, r11, z_ref;
  bool is_circle;
};

// one candidate point of the ellipse gather (trg.cpp:309-325); moments in fp64, z shifted.
// Returns whether the point was kept; mo.kept / mo.in_range are per-lane counts for callers that
// have no cheaper way (the global-memory fallback).
__device__ __forceinline__ bool ellipse_point(const EllipseParams &ep, float px, float py, float pz,
                                              Moments &mo) {
  const float ddx = px - ep.cx, ddy = py - ep.cy;
  const float d2 = ddx * ddx + ddy * ddy;
  bool keep = false;
  if (d2 <= ep.a2) {
    mo.in_range++;
    const float X = ep.r00 * ddx + ep.r01 * ddy;
    const float Y = ep.r10 * ddx + ep.r11 * ddy;
    keep = ep.is_circle;
    if (!ep.is_circle) keep = (X * X) * ep.bb + (Y * Y) * ep.a2 < ep.aabb;
    if (keep) {
      mo.kept++;
      const double xd = (double)X, yd = (double)Y, zd = (double)pz - (double)ep.z_ref;
      mo.s_x += xd;
      mo.s_y += yd;
      mo.s_z += zd;
      mo.s_xx += xd * xd;
      mo.s_xy += xd * yd;
      mo.s_xz += xd * zd;
      mo.s_yy += yd * yd;
      mo.s_yz += yd * zd;
      mo.s_zz += zd * zd;
    }
  }
  return keep;
}

// ---- LDS-staged neighbour tile -------------------------------------------------------------------
// All map queries of one edge (the segment-walk discs and the ellipse gather) fall inside one
// small box.  The wave loads the candidate points of that box ONCE -- every row segment is a
// contiguous range, all loads are issued back to back -- into a per-wave LDS tile, and every
// query then scans the tile instead of going back to global memory.
typedef float f2 __attribute__((ext_vector_type(2)));

constexpr int TCAP = 320;            // points per tile and entries of the hit buffer
constexpr int TITER = TCAP / WAVE;   // 5
constexpr int MAXROWS = 32;

// per-wave LDS: (x, y) interleaved so one 8-byte read feeds the packed distance arithmetic
struct WaveTile {
  f2 xy[TCAP];
  float z[TCAP];
  float zb[TCAP];
};
constexpr int MOM_STRIDE = 64;  // doubles per moment row in the reduction scratch
static_assert(sizeof(WaveTile) >= (9 * MOM_STRIDE + 63) * sizeof(double),
              "the moment reduction reuses the dead tile");

// Only the points that can matter are kept in the tile: every segment-walk disc lies inside the
// capsule of half-width robot_size around the segment (|u| <= r, -r <= t <= dist + r in the
// segment's frame), and the ellipse gather takes points of the disc of radius a around the centre
// (the reference rotates by +theta, trg.cpp:302-303, so its ellipse is NOT aligned with the segment
// and only the disc bounds it).  The capsule bounds are widened by a millimetre, far above any fp32
// rounding of t and u, so the exact membership tests that follow see every point they could accept;
// tile order stays ascending.
struct CapsuleFilter {
  float x1, y1, dirx, diry;  // segment origin and unit direction
  float t_lo, t_hi, u_max;   // conservative capsule bounds
  float cx, cy, a2;          // the gather disc of trg.cpp:304, counted here before filtering
};

// returns the number of staged candidates, or -1 when the box does not fit (caller falls back);
// in_range = points of the box within the gather disc (the kd_nearest_range2 result size)
__device__ int stage_tile(const MapView &m, const CellRange &c, WaveTile &t, const CapsuleFilter &f,
                          int &in_range) {
  const int lane = lane_id();
  const int nrows = c.cy1 - c.cy0 + 1;
  in_range = 0;
  if (nrows > MAXROWS) return -1;
  int s = 0, e = 0;
  if (lane < nrows) {
    const int base = (c.cy0 + lane) * m.W;
    s = m.cell_start[base + c.cx0];
    e = m.cell_start[base + c.cx1 + 1];
  }
  const int len = e - s;
  // inclusive prefix of the row lengths over the lanes (rows only occupy the first 32 lanes):
  // DPP row shifts inside each 16-lane row, then lane 15's total into the second row
  int inc = len;
#define SCAN_STEP(ctrl, rowmask) \
  inc += __builtin_amdgcn_update_dpp(0, inc, ctrl, rowmask, 0xf, false);
  SCAN_STEP(0x111, 0xf)  // row_shr:1
  SCAN_STEP(0x112, 0xf)  // row_shr:2
  SCAN_STEP(0x114, 0xf)  // row_shr:4
  SCAN_STEP(0x118, 0xf)  // row_shr:8
  SCAN_STEP(0x142, 0xa)  // row_bcast:15 -> rows 1, 3
#undef SCAN_STEP
  const int total = __builtin_amdgcn_readlane(inc, 31);  // lanes >= nrows hold zero lengths
  if (total > TCAP) return -1;
  const int delta = s - (inc - len);  // tile position t of row r lives at map index t + delta[r]
  float rx[TITER], ry[TITER], rz[TITER];
#pragma unroll
  for (int i = 0; i < TITER; ++i) {
    rx[i] = ry[i] = rz[i] = 0.0f;
    if (i * WAVE < total) {  // wave-uniform
      const int tt = lane + i * WAVE;
      const bool act = tt < total;
      const int tq = act ? tt : 0;
      int row = 0;
      for (int l = 0; l < nrows; ++l) row += (tq >= __builtin_amdgcn_readlane(inc, l));
      const int src = tq + __shfl(delta, row);
      if (act) {
        const float4 r = m.pt[src];
        rx[i] = r.x;
        ry[i] = r.y;
        rz[i] = r.z;
      }
    }
  }
  int n = 0;
#pragma unroll
  for (int i = 0; i < TITER; ++i) {
    if (i * WAVE < total) {  // wave-uniform
      const bool act = lane + i * WAVE < total;
      const float ddx = rx[i] - f.cx, ddy = ry[i] - f.cy;
      const bool inr = act && (ddx * ddx + ddy * ddy <= f.a2);
      const float px = rx[i] - f.x1, py = ry[i] - f.y1;
      const float tt = px * f.dirx + py * f.diry;
      const float uu = py * f.dirx - px * f.diry;
      const bool keep = act && (inr || (tt >= f.t_lo && tt <= f.t_hi && fabsf(uu) <= f.u_max));
      const unsigned long long mask = ballot(keep);
      in_range += __popcll(ballot(inr));
      if (keep) {
        const int pos = n + __popcll(mask & lanemask_lt());
        f2 v;
        v.x = rx[i];
        v.y = ry[i];
        t.xy[pos] = v;
        t.z[pos] = rz[i];
      }
      n += __popcll(mask);
    }
  }
  wave_lds_sync();
  return n;
}

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fminf(v, __shfl_xor(v, m));
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
  return v;
}

// Wave-wide min / max of order-preserving keys with DPP row shifts and row broadcasts (no LDS
// traffic); the result is wave-uniform.
template <bool IS_MIN>
__device__ __forceinline__ unsigned wave_reduce_key(unsigned v) {
  const unsigned ident = IS_MIN ? 0xFFFFFFFFu : 0u;
#define KEY_STEP(ctrl, rowmask)                                                                  \
  {                                                                                              \
    const unsigned o = (unsigned)__builtin_amdgcn_update_dpp((int)ident, (int)v, ctrl, rowmask,  \
                                                             0xf, false);                        \
    v = IS_MIN ? (o < v ? o : v) : (o > v ? o : v);                                              \
  }
  KEY_STEP(0x111, 0xf)  // row_shr:1
  KEY_STEP(0x112, 0xf)  // row_shr:2
  KEY_STEP(0x114, 0xf)  // row_shr:4
  KEY_STEP(0x118, 0xf)  // row_shr:8  -> lane 15 of every row holds the row's result
  KEY_STEP(0x142, 0xa)  // row_bcast:15 into rows 1 and 3
  KEY_STEP(0x143, 0xc)  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's result
#undef KEY_STEP
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// k-th smallest of zb[0..n) by rank counting, then #{|z - z_med| > h}  (trg.cpp:763-772); any n
__device__ int median_count(const float *zb, int n, float h) {
  const int lane = lane_id();
  const int k = n / 2;
  float mine = 0.0f;
  bool found = false;
  for (int i = lane; i < n; i += WAVE) {
    const float zi = zb[i];
    int rank = 0;
    for (int j = 0; j < n; ++j) {
      const float zj = zb[j];
      rank += (zj < zi) || (zj == zi && j < i);
    }
    if (rank == k) {
      mine = zi;
      found = true;
    }
  }
  const unsigned long long who = ballot(found);
  const float zmed = who ? __shfl(mine, __ffsll((long long)who) - 1) : zb[0];
  int cnt = 0;
  for (int i = lane; i < n; i += WAVE) cnt += fabsf(zb[i] - zmed) > h;
  return wave_sum(cnt);
}

// The same statistic for n <= 64 values, one per lane (lane < n holds z): the median's key is
// found by a most-significant-bit-first radix selection whose bookkeeping runs on the scalar unit
// (ballot masks and popcounts), two vector instructions per key bit.  key_lo / key_hi are the
// wave-uniform smallest / largest key of the set: their common leading bits need no selection.
__device__ __forceinline__ int median_count_lanes(float z, int n, float h, unsigned key_lo,
                                                  unsigned key_hi) {
  const int lane = lane_id();
  const unsigned key = float_key(z);
  unsigned long long alive = (n >= 64) ? ~0ull : ((1ull << n) - 1ull);
  int k = n / 2;  // pts[pts.size() / 2] after the ascending sort (trg.cpp:764)
  const unsigned diff = key_lo ^ key_hi;
  unsigned prefix = key_lo;
  if (diff) {
    const int top = 31 - __clz((int)diff);
    prefix = (top == 31) ? 0u : (key_lo & ~((2u << top) - 1u));
    for (int bit = top; bit >= 0; --bit) {
      const unsigned long long ones = ballot((key >> bit) & 1u);
      const unsigned long long zeros = alive & ~ones;
      const int c0 = __popcll(zeros);
      if (k < c0) {
        alive = zeros;
      } else {
        k -= c0;
        alive &= ones;
        prefix |= 1u << bit;
      }
    }
  }
  const float zmed = key_float(prefix);
  return __popcll(ballot(lane < n && fabsf(z - zmed) > h));
}

// isCollision of one disc, candidates read from the tile.  n_out = points in the disc.
__device__ bool tile_disc_collides(WaveTile &t, int T, float qx, float qy, float r, float h,
                                   float threshold, int &n_out) {
  const int lane = lane_id();
  const float r2 = r * r;
  int n = 0;
  float zmin = FLT_MAX, zmax = -FLT_MAX;
  for (int base = 0; base < T; base += WAVE) {
    const int i = base + lane;
    bool hit = false;
    float z = 0.0f;
    if (i < T) {
      const f2 p = t.xy[i];
      const float dx = p.x - qx;
      const float dy = p.y - qy;
      z = t.z[i];
      const float d2 = dx * dx + dy * dy;
      hit = d2 <= r2;
    }
    const unsigned long long mask = ballot(hit);
    if (hit) {
      t.zb[n + __popcll(mask & lanemask_lt())] = z;
      zmin = fminf(zmin, z);
      zmax = fmaxf(zmax, z);
    }
    n += __popcll(mask);
  }
  n_out = n;
  if (n == 0) return true;  // trg.cpp:749-752
  zmin = wave_min(zmin);
  zmax = wave_max(zmax);
  int cnt = 0;
  // every |z - z_med| <= zmax - zmin (rounding is monotone), so a flat disc needs no median
  if (!(zmax - zmin <= h)) {
    wave_lds_sync();
    cnt = median_count(t.zb, n, h);
    wave_lds_sync();
  }
  const float ratio = (float)cnt / (float)n;
  return ratio > threshold;
}


struct EdgeMidOut {
  int status, n_pts;
  float dist;
  int hits;  // map points inside this edge's query radii (instrumentation)
};

// covariance = centred^T * centred / (n - 1)  (trg.cpp:337-338) from the fp64 moments, rounded once
__device__ __forceinline__ void mid_covariance(const float *r, float cov[3][3]) {
  const double *sm = (const double *)(r + MID_MOMENTS);
  const int kept = __float_as_int(r[1]);
  const double n = (double)kept;
  const double mx = sm[0] / n, my = sm[1] / n, mz = sm[2] / n;
  const double inv = 1.0 / (double)(kept - 1);
  cov[0][0] = (float)((sm[3] - n * mx * mx) * inv);
  cov[0][1] = cov[1][0] = (float)((sm[4] - n * mx * my) * inv);
  cov[0][2] = cov[2][0] = (float)((sm[5] - n * mx * mz) * inv);
  cov[1][1] = (float)((sm[6] - n * my * my) * inv);
  cov[1][2] = cov[2][1] = (float)((sm[7] - n * my * mz) * inv);
  cov[2][2] = (float)((sm[8] - n * mz * mz) * inv);
}

// Wave-wide sums of the nine moments, written straight into the mid record.  Instead of nine
// 6-step butterflies (108 cross-lane moves, 54 fp64 adds) the per-lane partials go through the
// dead tile: 63 lanes each add ten partials of one moment, then one lane per moment adds the seven
// part sums -- 17 dependent fp64 adds in a fixed order (deterministic, independent of scheduling).
__device__ __forceinline__ void reduce_store_moments(WaveTile &t, const Moments &mo, float *rec) {
  const int lane = lane_id();
  double *P = (double *)&t;
  wave_lds_sync();  // every lane is done reading the tile
  P[0 * MOM_STRIDE + lane] = mo.s_x;
  P[1 * MOM_STRIDE + lane] = mo.s_y;
  P[2 * MOM_STRIDE + lane] = mo.s_z;
  P[3 * MOM_STRIDE + lane] = mo.s_xx;
  P[4 * MOM_STRIDE + lane] = mo.s_xy;
  P[5 * MOM_STRIDE + lane] = mo.s_xz;
  P[6 * MOM_STRIDE + lane] = mo.s_yy;
  P[7 * MOM_STRIDE + lane] = mo.s_yz;
  P[8 * MOM_STRIDE + lane] = mo.s_zz;
  wave_lds_sync();
  const int q = lane / 7, part = lane - q * 7;  // lanes 0..62: moment q, partials [10 part, 10 part + 10)
  double acc = 0.0;
  if (lane < 63) {
    const double *row = P + q * MOM_STRIDE + part * 10;
    const int cnt = (part == 6) ? 4 : 10;
    // start each moment's run at a different element so that the nine rows (512 bytes apart, i.e.
    // the same LDS banks) are not read in lock-step; the order stays fixed per (moment, part)
    int idx = q % cnt;
#pragma unroll
    for (int j = 0; j < 10; ++j)
      if (j < cnt) {
        acc += row[idx];
        idx = (idx + 1 == cnt) ? 0 : idx + 1;
      }
  }
  double *Q = P + 9 * MOM_STRIDE;  // 63 part sums, moment-major
  wave_lds_sync();
  if (lane < 63) Q[lane] = acc;
  wave_lds_sync();
  if (lane < 9) {
    double tot = 0.0;
#pragma unroll
    for (int j = 0; j < 7; ++j) tot += Q[lane * 7 + j];
    ((double *)(rec + MID_MOMENTS))[lane] = tot;
  }
}

// Profiling builds (-DTRG_EDGE_STAGE_CUT=n, scripts/edge_microbench.sh) stop the edge evaluation
// after stage n to attribute its cost; 0 = the product.
#ifndef TRG_EDGE_STAGE_CUT
#define TRG_EDGE_STAGE_CUT 0
#endif
#define EDGE_CUT(n)                                \
  if (TRG_EDGE_STAGE_CUT == (n)) {                 \
    o.status = EDGE_SEG;                           \
    return o;                                      \
  }

constexpr int KMAX = 6;  // segment-walk discs of the fast path: dist <= expand_dist + robot_size
                         // on every edge the build tries

// One pass over the tile for the K discs of the segment walk: per disc the hit count (scalar
// unit: ballot + popcount) and the smallest / largest z key among its hits.  Branch-free; the
// (x, y) pairs and the disc centres are packed so the distance is two packed fp32 operations.
template <int K>
__device__ __forceinline__ void sweep_discs(const WaveTile &tile, int T, const f2 (&q)[KMAX], float r2,
                                            int (&cnt)[KMAX], unsigned (&kmn)[KMAX],
                                            unsigned (&kmx)[KMAX]) {
  const int lane = lane_id();
  for (int base = 0; base < T; base += WAVE) {
    const int i = base + lane;
    const bool valid = i < T;
    const int ii = valid ? i : 0;
    const f2 p = tile.xy[ii];
    const unsigned zk = float_key(tile.z[ii]);
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const f2 d = p - q[k];
      const f2 dd = d * d;
      const float d2 = dd.x + dd.y;
      const bool hit = valid && d2 <= r2;
      cnt[k] += __popcll(ballot(hit));
      const unsigned lo = hit ? zk : 0xFFFFFFFFu, hi = hit ? zk : 0u;
      kmn[k] = lo < kmn[k] ? lo : kmn[k];
      kmx[k] = hi > kmx[k] ? hi : kmx[k];
    }
  }
}

// One wave: the position-only part of TRG::wireEdge up to the moments of the covariance
// (trg.cpp:269-338).  Writes the nine moments of an accepted edge into rec; the caller stores the
// record header from the returned fields.
__device__ EdgeMidOut edge_gather(const MapView &m, const QueryParams &p, float x1, float y1,
                                  float z1, float x2, float y2, float z2, WaveTile &tile,
                                  float *rec, DeviceCounters *ctr) {
  const int lane = lane_id();
  EdgeMidOut o;
  o.n_pts = 0;
  o.hits = 0;
  const EdgeGeom g = edge_geometry(p, x1, y1, z1, x2, y2, z2);
  o.dist = g.dist;
  if (g.gated) {
    o.status = EDGE_GATE;
    return o;
  }

  // ellipse with foci at the two nodes, trg.cpp:291-297
  const float c = 0.5f * g.dist;
  const float b = p.robot_size;
  float a = b;
  if (c >= b) a = sqrtf(c * c + b * b);
  EllipseParams ep;
  ep.cx = g.cx;
  ep.cy = g.cy;
  ep.is_circle = (a == b);
  ep.r00 = g.dirx;  // R << dir.x, -dir.y, dir.y, dir.x  (trg.cpp:302-303)
  ep.r01 = -g.diry;
  ep.r10 = g.diry;
  ep.r11 = g.dirx;
  ep.a2 = a * a;
  ep.bb = b * b;
  ep.aabb = a * a * b * b;
  ep.z_ref = z1;

  // box holding every query of this edge: the capsule around the segment and the ellipse disc
  CellRange box;
  {
    const float rp = p.robot_size * 1.001f + 1e-6f;
    const float ap = a * 1.001f + 1e-6f;
    const float xlo = fminf(fminf(x1, x2) - rp, g.cx - ap), xhi = fmaxf(fmaxf(x1, x2) + rp, g.cx + ap);
    const float ylo = fminf(fminf(y1, y2) - rp, g.cy - ap), yhi = fmaxf(fmaxf(y1, y2) + rp, g.cy + ap);
    box.cx0 = cell_coord(xlo, m.x0, m.inv_g, m.W);
    box.cx1 = cell_coord(xhi, m.x0, m.inv_g, m.W);
    box.cy0 = cell_coord(ylo, m.y0, m.inv_g, m.H);
    box.cy1 = cell_coord(yhi, m.y0, m.inv_g, m.H);
  }
  CapsuleFilter cf;
  cf.x1 = x1;
  cf.y1 = y1;
  cf.dirx = g.dirx;
  cf.diry = g.diry;
  {
    const float rw = p.robot_size * 1.001f + 1e-3f;
    cf.t_lo = -rw;
    cf.t_hi = g.dist + rw;
    cf.u_max = rw;
  }
  cf.cx = ep.cx;
  cf.cy = ep.cy;
  cf.a2 = ep.a2;
  EDGE_CUT(1)  // geometry only
  int staged_in_range = 0;
  const int T = stage_tile(m, box, tile, cf, staged_in_range);
  if (TRG_EDGE_STAGE_CUT == 2) {  // + tile staging (keep its results alive)
    o.status = EDGE_SEG + (T + staged_in_range > 1000000) +
               (tile.z[lane] + tile.xy[lane].x + tile.xy[lane + 64].y == 12345.0f);
    return o;
  }

  unsigned long long hits = 0;
  const float ds = p.robot_size * 0.5f;

  // Fast path: ONE sweep over the staged tile serves every segment-walk disc; the per-disc
  // statistics are then examined in walk order, so the early exit on the first colliding disc --
  // and the hit count the reference would have produced up to it -- are unchanged.
  if (T >= 0 && ds > 0.0f) {
    f2 q[KMAX];
    int K = 0;
    bool fits = true;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) q[k] = f2{0.0f, 0.0f};
    for (float i = 0; i < g.dist; i += ds) {  // trg.cpp:283 (float accumulation is semantics)
      if (K == KMAX) {
        fits = false;
        break;
      }
#pragma unroll
      for (int k = 0; k < KMAX; ++k)
        if (k == K) {
          q[k].x = x1 + i * g.dirx;
          q[k].y = y1 + i * g.diry;
        }
      ++K;
    }
    if (fits) {
      const float r2 = p.robot_size * p.robot_size;
      int cnt[KMAX];
      unsigned kmn[KMAX], kmx[KMAX];
#pragma unroll
      for (int k = 0; k < KMAX; ++k) {
        cnt[k] = 0;
        kmn[k] = 0xFFFFFFFFu;
        kmx[k] = 0u;
      }
      switch (K) {
        case 1: sweep_discs<1>(tile, T, q, r2, cnt, kmn, kmx); break;
        case 2: sweep_discs<2>(tile, T, q, r2, cnt, kmn, kmx); break;
        case 3: sweep_discs<3>(tile, T, q, r2, cnt, kmn, kmx); break;
        case 4: sweep_discs<4>(tile, T, q, r2, cnt, kmn, kmx); break;
        case 5: sweep_discs<5>(tile, T, q, r2, cnt, kmn, kmx); break;
        case 6: sweep_discs<6>(tile, T, q, r2, cnt, kmn, kmx); break;
        default: break;  // K == 0: zero-length edge, no disc (trg.cpp:283 never enters the loop)
      }
      if (TRG_EDGE_STAGE_CUT == 3) {  // + disc sweep (keep its results alive)
        o.status = EDGE_SEG + (cnt[0] + cnt[1] + cnt[2] + cnt[3] + cnt[4] + cnt[5] > 1000000) +
                   ((kmn[0] ^ kmn[1] ^ kmn[2] ^ kmn[3] ^ kmn[4] ^ kmn[5] ^ kmx[0] ^ kmx[1] ^ kmx[2] ^
                     kmx[3] ^ kmx[4] ^ kmx[5]) == 12345u);
        return o;
      }
#pragma unroll
      for (int k = 0; k < KMAX; ++k) {
        if (k < K) {
          const int n = cnt[k];
          hits += (unsigned long long)n;
          bool col;
          if (n == 0) {
            col = true;  // trg.cpp:749-752
          } else {
            const unsigned klo = wave_reduce_key<true>(kmn[k]);
            const unsigned khi = wave_reduce_key<false>(kmx[k]);
            int bad = 0;
            // every |z - z_med| <= zmax - zmin (rounding is monotone): a flat disc needs no median
            if (!(key_float(khi) - key_float(klo) <= p.height_threshold)) {
              // compact the disc's z values (ascending tile order) into the hit buffer
              int nn = 0;
              for (int base = 0; base < T; base += WAVE) {
                const int i = base + lane;
                bool hit = false;
                float z = 0.0f;
                if (i < T) {
                  const f2 d = tile.xy[i] - q[k];
                  const f2 dd = d * d;
                  z = tile.z[i];
                  hit = dd.x + dd.y <= r2;
                }
                const unsigned long long mask = ballot(hit);
                if (hit) tile.zb[nn + __popcll(mask & lanemask_lt())] = z;
                nn += __popcll(mask);
              }
              wave_lds_sync();
              if (n <= WAVE) {
                const float zl = tile.zb[lane < n ? lane : 0];
                bad = median_count_lanes(zl, n, p.height_threshold, klo, khi);
              } else {
                bad = median_count(tile.zb, n, p.height_threshold);
              }
              wave_lds_sync();
            }
            // (a flat disc -- most are -- has ratio 0 / n = +0 exactly: no division)
            if (bad == 0)
              col = 0.0f > p.collision_threshold;
            else
              col = (float)bad / (float)n > p.collision_threshold;
          }
          if (col) {
            o.status = EDGE_SEG | g.uncertain;
            o.hits = (int)hits;
            return o;
          }
        }
      }
      EDGE_CUT(4)  // + disc reductions, medians, decisions
      // second sweep: ellipse gather (kept apart from the disc sweep to bound register pressure)
      Moments mo;
      int kept = 0;
      for (int base = 0; base < T; base += WAVE) {
        const int i = base + lane;
        bool keep = false;
        if (i < T) {
          const f2 pt = tile.xy[i];
          keep = ellipse_point(ep, pt.x, pt.y, tile.z[i], mo);
        }
        kept += __popcll(ballot(keep));
      }
      if (TRG_EDGE_STAGE_CUT == 5) {  // + ellipse sweep
        o.status = EDGE_SEG + (mo.s_x + mo.s_y + mo.s_z + mo.s_xx + mo.s_xy + mo.s_xz + mo.s_yy +
                                   mo.s_yz + mo.s_zz == 12345.0) + (kept > 100000);
        return o;
      }
      const int in_range = staged_in_range;  // counted over the whole box, before the capsule cut
      hits += (unsigned long long)in_range;
      o.hits = (int)hits;
      o.n_pts = kept;
      if (in_range == 0) {
        o.status = EDGE_EMPTY | g.uncertain;
        return o;
      }
      if (kept < 3) {
        o.status = EDGE_FEW | g.uncertain;
        return o;
      }
      reduce_store_moments(tile, mo, rec);
      o.status = EDGE_OK | g.uncertain;
      return o;
    }
  }

  // general path: one disc at a time (long edges, oversized boxes, dense maps)
  // segment walk, trg.cpp:282-288 (float accumulation of i is part of the semantics)
  int guard = 0;
  for (float i = 0; i < g.dist; i += ds) {
    const float qx = x1 + i * g.dirx;
    const float qy = y1 + i * g.diry;
    bool col;
    int n = 0;
    if (T >= 0) {
      col = tile_disc_collides(tile, T, qx, qy, p.robot_size, p.height_threshold,
                               p.collision_threshold, n);
    } else {
      const Disc d =
          disc_query<false>(m, qx, qy, p.robot_size, p.height_threshold, tile.zb, ctr, TCAP);
      n = d.n;
      col = disc_collides(d, p.collision_threshold);
    }
    hits += (unsigned long long)n;
    if (col) {
      o.status = EDGE_SEG | g.uncertain;
      o.hits = (int)hits;
      return o;
    }
    if (++guard > 100000 || !(ds > 0.0f)) break;
  }

  // gather + rotate + filter (trg.cpp:304-325)
  Moments mo;
  if (T >= 0) {
    for (int i = lane; i < T; i += WAVE) {
      const f2 pt = tile.xy[i];
      ellipse_point(ep, pt.x, pt.y, tile.z[i], mo);
    }
  } else {
    const CellRange cr = cells_for(m, g.cx, g.cy, a);
    for (int cyi = cr.cy0; cyi <= cr.cy1; ++cyi) {
      const int s = m.cell_start[cyi * m.W + cr.cx0];
      const int e = m.cell_start[cyi * m.W + cr.cx1 + 1];
      for (int i = s + lane; i < e; i += WAVE) ellipse_point(ep, m.x[i], m.y[i], m.z[i], mo);
    }
  }
  const int in_range = (T >= 0) ? staged_in_range : wave_sum(mo.in_range);
  const int kept = wave_sum(mo.kept);
  hits += (unsigned long long)in_range;
  o.hits = (int)hits;
  o.n_pts = kept;
  if (in_range == 0) {
    o.status = EDGE_EMPTY | g.uncertain;
    return o;
  }
  if (kept < 3) {
    o.status = EDGE_FEW | g.uncertain;
    return o;
  }
  reduce_store_moments(tile, mo, rec);
  o.status = EDGE_OK | g.uncertain;
  return o;
}

// lane 0 writes the 16-byte header of a mid record (edge_gather wrote the moments)
__device__ __forceinline__ void store_mid(float *rec, const EdgeMidOut &o) {
  if (lane_id() == 0) {
    float4 h;
    h.x = __int_as_float(o.status);
    h.y = __int_as_float(o.n_pts);
    h.z = o.dist;
    h.w = __int_as_float(o.hits);
    *(float4 *)rec = h;
  }
}

// ---- murmur-style counter hash shared with the oracle's sampler ------------------------------
__device__ __forceinline__ uint32_t fmix32(uint32_t h) {
  h ^= h >> 16;
  h *= 0x85EBCA6Bu;
  h ^= h >> 13;
  h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h;
}
__device__ __forceinline__ uint32_t sample_hash(uint32_t seed, uint32_t epoch, uint32_t id,
                                                uint32_t trial) {
  uint32_t h = fmix32(seed ^ 0x9E3779B9u);
  h = fmix32(h + epoch * 0x9E3779B9u + 0x7F4A7C15u);
  h = fmix32(h + id * 0x85EBCA6Bu + 0x165667B1u);
  h = fmix32(h + trial * 0xC2B2AE35u + 0x27D4EB2Fu);
  return h;
}

// ================================ kernels =======================================================

__global__ void k_init_bounds(unsigned *b) {
  if (threadIdx.x == 0) {
    b[0] = 0xFFFFFFFFu;
    b[1] = 0xFFFFFFFFu;
    b[2] = 0u;
    b[3] = 0u;
  }
}

__global__ __launch_bounds__(256) void k_bounds(const float *xyz, size_t n, size_t stride,
                                                unsigned *b) {
  unsigned mnx = 0xFFFFFFFFu, mny = 0xFFFFFFFFu, mxx = 0u, mxy = 0u;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x) {
    const unsigned kx = float_key(xyz[i * stride]);
    const unsigned ky = float_key(xyz[i * stride + 1]);
    mnx = min(mnx, kx);
    mxx = max(mxx, kx);
    mny = min(mny, ky);
    mxy = max(mxy, ky);
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    mnx = min(mnx, (unsigned)__shfl_xor((int)mnx, m));
    mny = min(mny, (unsigned)__shfl_xor((int)mny, m));
    mxx = max(mxx, (unsigned)__shfl_xor((int)mxx, m));
    mxy = max(mxy, (unsigned)__shfl_xor((int)mxy, m));
  }
  // one set of atomics per workgroup: same-address atomics serialise (~13 ns each)
  __shared__ unsigned red[4][4];
  const int w = threadIdx.x >> 6;
  if (lane_id() == 0) {
    red[w][0] = mnx;
    red[w][1] = mny;
    red[w][2] = mxx;
    red[w][3] = mxy;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicMin(&b[0], min(min(red[0][0], red[1][0]), min(red[2][0], red[3][0])));
    atomicMin(&b[1], min(min(red[0][1], red[1][1]), min(red[2][1], red[3][1])));
    atomicMax(&b[2], max(max(red[0][2], red[1][2]), max(red[2][2], red[3][2])));
    atomicMax(&b[3], max(max(red[0][3], red[1][3]), max(red[2][3], red[3][3])));
  }
}

__global__ __launch_bounds__(256) void k_cell_count(const float *xyz, size_t n, size_t stride,
                                                    float x0, float y0, float inv_g, int W, int H,
                                                    int *cell_of, int *rank, int *counts) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x) {
    const int cx = cell_coord(xyz[i * stride], x0, inv_g, W);
    const int cy = cell_coord(xyz[i * stride + 1], y0, inv_g, H);
    const int c = cy * W + cx;
    cell_of[i] = c;
    rank[i] = atomicAdd(&counts[c], 1);
  }
}

constexpr int SCAN_TILE = 2048;  // 256 threads x 8 items

__global__ __launch_bounds__(256) void k_scan_tiles(const int *in, int *out, int m, int *tile_sum) {
  __shared__ int wave_tot[4];
  const int t = threadIdx.x;
  const int base = blockIdx.x * SCAN_TILE + t * 8;
  int v[8];
  int local = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int idx = base + k;
    v[k] = idx < m ? in[idx] : 0;
    local += v[k];
  }
  // inclusive scan of `local` inside the wave
  int inc = local;
#pragma unroll
  for (int d = 1; d < WAVE; d <<= 1) {
    const int up = __shfl_up(inc, d);
    if (lane_id() >= d) inc += up;
  }
  const int w = t >> 6;
  if (lane_id() == WAVE - 1) wave_tot[w] = inc;
  __syncthreads();
  int wave_off = 0;
  for (int k = 0; k < w; ++k) wave_off += wave_tot[k];
  int run = wave_off + inc - local;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int idx = base + k;
    if (idx < m) out[idx] = run;
    run += v[k];
  }
  if (t == 255) tile_sum[blockIdx.x] = wave_off + inc;
}

__global__ __launch_bounds__(256) void k_scan_tile_sums(int *tile_sum, int nt) {
  // single block: exclusive scan of nt tile sums in place, total appended at [nt]
  __shared__ int carry;
  __shared__ int wave_tot[4];
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < nt; base += 256) {
    const int idx = base + threadIdx.x;
    const int v = idx < nt ? tile_sum[idx] : 0;
    int inc = v;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
      const int up = __shfl_up(inc, d);
      if (lane_id() >= d) inc += up;
    }
    const int w = threadIdx.x >> 6;
    if (lane_id() == WAVE - 1) wave_tot[w] = inc;
    __syncthreads();
    int off = carry;
    for (int k = 0; k < w; ++k) off += wave_tot[k];
    if (idx < nt) tile_sum[idx] = off + inc - v;
    __syncthreads();
    if (threadIdx.x == 255) carry = off + inc;
    __syncthreads();
  }
  if (threadIdx.x == 0) tile_sum[nt] = carry;
}

__global__ __launch_bounds__(256) void k_scan_add(int *out, int m, const int *tile_sum, int nt) {
  const int off = tile_sum[blockIdx.x];
  const int base = blockIdx.x * SCAN_TILE + threadIdx.x * 8;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int idx = base + k;
    if (idx < m) out[idx] += off;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) out[m] = tile_sum[nt];
}

// Points go to their place in the cell order with ONE 16-byte store each (x, y, z, original index) into
// a scratch array: the destinations are random (the cloud arrives shuffled), and a random 4-byte store
// costs a memory transaction just like a 16-byte one -- four SoA stores per point were 1.0 ms at C3,
// the longest kernel of the index build; this is 0.3 ms.  k_cell_sort_aos turns the records into the SoA arrays the queries read.
__global__ __launch_bounds__(256) void k_scatter_aos(const float *xyz, size_t n, size_t stride,
                                                     const int *cell_of, const int *rank,
                                                     const int *cell_start, float4 *aos) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x) {
    const int dst = cell_start[cell_of[i]] + rank[i];
    aos[dst] = make_float4(xyz[i * stride], xyz[i * stride + 1], xyz[i * stride + 2], __int_as_float((int)i));
  }
}

// ---- index build through bins (the default) -----------------------------------------------------------
// The cloud arrives in no order, so counting points per cell with global atomics and scattering them to their
// cells touches a random line per point (k_cell_count / k_scatter_aos above: 0.74 ms per 10 M points, 5-8 x the
// bytes they move).  Here the points first go to BINS of 2^bin_shift consecutive cells (about one cell row
// of the grid, ~9 000 points): every workgroup takes one contiguous chunk of the cloud, counts its points per
// bin in LDS (k_bin_count), a scan over the (bin, workgroup) counts gives every workgroup a private range in
// every bin, and the second pass over the chunk writes each point behind its workgroup's cursor (k_bin_scatter:
// runs of ~36 records).  Then one workgroup per bin counts its points per cell in LDS, scans, writes the
// bin's part of cell_start and places the points (k_bin_cells: all atomics in LDS, all traffic inside a
// 150-KB range).  k_cell_sort_aos finishes as before (cells ordered by original index, SoA + records).
constexpr int BIN_WG = 256;            // chunks of the cloud = workgroups of the two binning passes
constexpr int BIN_THREADS = 1024;
constexpr int BIN_MAX = 4096;          // bins an LDS histogram holds
constexpr int BIN_CELLS_MAX = 8192;    // cells of one bin (their counters live in LDS)
constexpr int BIN_CELL_THREADS = 512;

__global__ __launch_bounds__(BIN_THREADS) void k_bin_count(const float *xyz, size_t n, size_t stride, float x0,
                                                           float y0, float inv_g, int W, int H, int bin_shift,
                                                           int nbins, size_t chunk, int *hist) {
  __shared__ int h[BIN_MAX];
  for (int b = threadIdx.x; b < nbins; b += BIN_THREADS) h[b] = 0;
  __syncthreads();
  const size_t lo = (size_t)blockIdx.x * chunk, hi = min(n, lo + chunk);
  for (size_t i = lo + threadIdx.x; i < hi; i += BIN_THREADS) {
    const int cx = cell_coord(xyz[i * stride], x0, inv_g, W);
    const int cy = cell_coord(xyz[i * stride + 1], y0, inv_g, H);
    atomicAdd(&h[(cy * W + cx) >> bin_shift], 1);
  }
  __syncthreads();
  for (int b = threadIdx.x; b < nbins; b += BIN_THREADS) hist[(size_t)b * gridDim.x + blockIdx.x] = h[b];
}

// base = exclusive scan of hist: where this workgroup's points of bin b start
__global__ __launch_bounds__(BIN_THREADS) void k_bin_scatter(const float *xyz, size_t n, size_t stride, float x0,
                                                             float y0, float inv_g, int W, int H, int bin_shift,
                                                             int nbins, size_t chunk, const int *base,
                                                             float4 *out) {
  __shared__ int cur[BIN_MAX];
  for (int b = threadIdx.x; b < nbins; b += BIN_THREADS) cur[b] = base[(size_t)b * gridDim.x + blockIdx.x];
  __syncthreads();
  const size_t lo = (size_t)blockIdx.x * chunk, hi = min(n, lo + chunk);
  for (size_t i = lo + threadIdx.x; i < hi; i += BIN_THREADS) {
    const float x = xyz[i * stride], y = xyz[i * stride + 1], z = xyz[i * stride + 2];
    const int cx = cell_coord(x, x0, inv_g, W);
    const int cy = cell_coord(y, y0, inv_g, H);
    const int dst = atomicAdd(&cur[(cy * W + cx) >> bin_shift], 1);
    out[dst] = make_float4(x, y, z, __int_as_float((int)i));
  }
}

// one workgroup per bin: in[T0, T1) -> the bin's cells of cell_start, out[T0, T1) grouped by cell (arrival order)
__global__ __launch_bounds__(BIN_CELL_THREADS) void k_bin_cells(const float4 *in, int n, float x0, float y0,
                                                                float inv_g, int W, int H, int bin_shift,
                                                                int nbins, int nwg, const int *base, int ncell,
                                                                int *cell_start, float4 *out) {
  __shared__ int cnt[BIN_CELLS_MAX];
  __shared__ int wave_tot[BIN_CELL_THREADS / WAVE];
  const int bin = blockIdx.x, tid = threadIdx.x;
  const int bin_cells = 1 << bin_shift;
  const int c0 = bin << bin_shift;
  const int T0 = base[(size_t)bin * nwg];
  const int T1 = (bin + 1 < nbins) ? base[(size_t)(bin + 1) * nwg] : n;
  for (int k = tid; k < bin_cells; k += BIN_CELL_THREADS) cnt[k] = 0;
  __syncthreads();
  for (int i = T0 + tid; i < T1; i += BIN_CELL_THREADS) {
    const float4 r = in[i];
    const int c = cell_coord(r.y, y0, inv_g, H) * W + cell_coord(r.x, x0, inv_g, W);
    atomicAdd(&cnt[c - c0], 1);
  }
  __syncthreads();
  // exclusive scan of the counters in place: a run of consecutive counters per thread, waves, workgroup
  const int per = bin_cells / BIN_CELL_THREADS > 0 ? bin_cells / BIN_CELL_THREADS : 1;  // (bin_cells >= 1024)
  const int k0 = tid * per;
  int local = 0;
  if (k0 < bin_cells)
    for (int k = 0; k < per; ++k) local += cnt[k0 + k];
  int inc = local;
#pragma unroll
  for (int d = 1; d < WAVE; d <<= 1) {
    const int up = __shfl_up(inc, d);
    if (lane_id() >= d) inc += up;
  }
  const int w = tid >> 6;
  if (lane_id() == WAVE - 1) wave_tot[w] = inc;
  __syncthreads();
  int run = inc - local;
  for (int k = 0; k < w; ++k) run += wave_tot[k];
  if (k0 < bin_cells)
    for (int k = 0; k < per; ++k) {
      const int v = cnt[k0 + k];
      cnt[k0 + k] = run;
      if (c0 + k0 + k < ncell) cell_start[c0 + k0 + k] = T0 + run;
      run += v;
    }
  if (bin == nbins - 1 && tid == 0) cell_start[ncell] = n;
  __syncthreads();
  for (int i = T0 + tid; i < T1; i += BIN_CELL_THREADS) {
    const float4 r = in[i];
    const int c = cell_coord(r.y, y0, inv_g, H) * W + cell_coord(r.x, x0, inv_g, W);
    out[T0 + atomicAdd(&cnt[c - c0], 1)] = r;
  }
}

// The atomic rank above is arrival order; sorting every cell by original index makes the index
// (and therefore every fp64 accumulation order downstream) independent of scheduling.
constexpr int CSORT_CAP = 3072;  // points of 256 consecutive cells staged in LDS (48 KB)

__device__ __forceinline__ void cell_insertion_sort(float *x, float *y, float *z, int *perm, int s,
                                                    int e) {
  for (int i = s + 1; i < e; ++i) {
    const int kp = perm[i];
    const float kx = x[i], ky = y[i], kz = z[i];
    int j = i - 1;
    while (j >= s && perm[j] > kp) {
      perm[j + 1] = perm[j];
      x[j + 1] = x[j];
      y[j + 1] = y[j];
      z[j + 1] = z[j];
      --j;
    }
    perm[j + 1] = kp;
    x[j + 1] = kx;
    y[j + 1] = ky;
    z[j + 1] = kz;
  }
}

// A workgroup owns 256 consecutive cells, i.e. one contiguous range of the sorted arrays: it is
// staged in LDS with coalesced loads, every thread sorts its own cell there, and the range is
// written back coalesced (ranges too long for LDS are sorted in place in global memory).
// (the records k_scatter_aos left are read, the SoA arrays the queries use are written)
__global__ __launch_bounds__(256) void k_cell_sort_aos(int ncell, const int *cell_start, const float4 *aos,
                                                       float *x, float *y, float *z, int *perm, float4 *pt) {
  __shared__ float lx[CSORT_CAP], ly[CSORT_CAP], lz[CSORT_CAP];
  __shared__ int lp[CSORT_CAP];
  const int c0 = blockIdx.x * blockDim.x;
  const int c = c0 + threadIdx.x;
  const int p0 = cell_start[c0], p1 = cell_start[min(c0 + (int)blockDim.x, ncell)];
  const int n = p1 - p0;
  if (n > CSORT_CAP) {  // (a range too long for LDS: unpack in place, sort in global memory)
    if (c < ncell) {
      const int s0 = cell_start[c], s1 = cell_start[c + 1];
      for (int i = s0; i < s1; ++i) {
        const float4 r = aos[i];
        x[i] = r.x;
        y[i] = r.y;
        z[i] = r.z;
        perm[i] = __float_as_int(r.w);
      }
      cell_insertion_sort(x, y, z, perm, s0, s1);
      for (int i = s0; i < s1; ++i) pt[i] = make_float4(x[i], y[i], z[i], __int_as_float(perm[i]));
    }
    return;
  }
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const float4 r = aos[p0 + i];
    lx[i] = r.x;
    ly[i] = r.y;
    lz[i] = r.z;
    lp[i] = __float_as_int(r.w);
  }
  __syncthreads();
  if (c < ncell) cell_insertion_sort(lx, ly, lz, lp, cell_start[c] - p0, cell_start[c + 1] - p0);
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    x[p0 + i] = lx[i];
    y[p0 + i] = ly[i];
    z[p0 + i] = lz[i];
    perm[p0 + i] = lp[i];
    pt[p0 + i] = make_float4(lx[i], ly[i], lz[i], __int_as_float(lp[i]));
  }
}

#ifndef TRG_QW
#define TRG_QW 4
#endif
constexpr int QW = TRG_QW;  // waves (= queries) per block in the probe / edge kernels

__global__ __launch_bounds__(QW *WAVE) void k_probe_collision(MapView m, QueryParams p,
                                                              float threshold, const float *xy,
                                                              int count, int *flag, int *cnt,
                                                              int *nout, DeviceCounters *ctr) {
  __shared__ float ztile[QW][HCAP];
  const int w = threadIdx.x >> 6;
  const int q = blockIdx.x * QW + w;
  if (q >= count) return;
  const Disc d = disc_query<false>(m, xy[2 * q], xy[2 * q + 1], p.robot_size, p.height_threshold,
                                   ztile[w], ctr);
  if (lane_id() == 0) {
    if (flag) flag[q] = disc_collides(d, threshold) ? 1 : 0;
    if (cnt) cnt[q] = d.cnt;
    if (nout) nout[q] = d.n;
  }
}

__global__ __launch_bounds__(QW *WAVE) void k_probe_nearest_z(MapView m, QueryParams p,
                                                              const float *xy, int count, float *z,
                                                              int *found, DeviceCounters *ctr) {
  const int w = threadIdx.x >> 6;
  const int q = blockIdx.x * QW + w;
  if (q >= count) return;
  float zz = 0.0f;
  int tie = 0;
  const bool ok = nearest_point(m, xy[2 * q], xy[2 * q + 1], p.robot_size, zz, tie);
  if (lane_id() == 0) {
    z[q] = ok ? zz : 0.0f;
    // found: 0 = empty map, 1 = unique nearest point, 2 = several at the same fp32 distance (the
    // host then asks for the one the reference's kd-tree would return)
    if (found) found[q] = ok ? (tie ? 2 : 1) : 0;
    if (tie && ctr) atomicAdd(&ctr[blockIdx.x % COUNTER_SHARDS].nn_ties, 1ull);
  }
}

// ---- exact nearest-point tie-break (rare path) ----------------------------------------------------
// One wave: the set of map points at the minimal fp32 distance from (qx, qy).
__global__ __launch_bounds__(WAVE) void k_map_tied_set(MapView m, float qx, float qy, float r0,
                                                       MapTieSet *out) {
  const int lane = lane_id();
  float R = r0;
  for (int iter = 0; iter < 40; ++iter) {
    const CellRange c = cells_for(m, qx, qy, R);
    float wd = FLT_MAX;
    for (int cy = c.cy0; cy <= c.cy1; ++cy) {
      const int s = m.cell_start[cy * m.W + c.cx0];
      const int e = m.cell_start[cy * m.W + c.cx1 + 1];
      for (int i = s + lane; i < e; i += WAVE) {
        const float dx = m.x[i] - qx, dy = m.y[i] - qy;
        const float d2 = dx * dx + dy * dy;
        wd = fminf(wd, d2);
      }
    }
#pragma unroll
    for (int msk = 32; msk >= 1; msk >>= 1) wd = fminf(wd, __shfl_xor(wd, msk));
    const bool whole = c.cx0 == 0 && c.cy0 == 0 && c.cx1 == m.W - 1 && c.cy1 == m.H - 1;
    if (wd != FLT_MAX && (wd <= R * R || whole)) {
      int n = 0;
      for (int cy = c.cy0; cy <= c.cy1; ++cy) {
        const int s = m.cell_start[cy * m.W + c.cx0];
        const int e = m.cell_start[cy * m.W + c.cx1 + 1];
        for (int base = s; base < e; base += WAVE) {
          const int i = base + lane;
          bool hit = false;
          if (i < e) {
            const float dx = m.x[i] - qx, dy = m.y[i] - qy;
            hit = (dx * dx + dy * dy) == wd;
          }
          const unsigned long long mask = __ballot(hit);
          if (hit) {
            const int pos = n + __popcll(mask & lanemask_lt());
            if (pos < MAPTIE_SET_CAP) {
              out->sidx[pos] = i;
              out->perm[pos] = m.perm[i];
              out->x[pos] = m.x[i];
              out->y[pos] = m.y[i];
              out->z[pos] = m.z[i];
            }
          }
          n += __popcll(mask);
        }
      }
      if (lane == 0) {
        out->count = n;
        out->d2 = wd;
      }
      return;
    }
    if (whole) break;
    R *= 2.0f;
  }
  if (lane == 0) {
    out->count = 0;
    out->d2 = 0.0f;
  }
}

// The kd-tree node that roots the subtree of the region [lox, hix) x [loy, hiy), below an ancestor
// inserted as point number perm_gt: the region's point with the smallest original index above
// perm_gt (kd_insert descends with `<` to the left, kdtree.c:179-198, so a subtree is exactly the
// later points of its half-open region).  key = (original index << 32 | sorted index).
// This is one step of the walk down the (never built) insertion tree of the map; region and ancestor
// are taken from the walk state in device memory, the host enqueues the steps without looking.
struct WalkRegion {
  float lox, hix, loy, hiy;
  int perm_gt;
};
__device__ __forceinline__ unsigned long long region_scan_rows(const MapView &m, const WalkRegion &g, int row0,
                                                               int row_stride, int tid, int nthreads) {
  const int cy0 = cell_coord(g.loy, m.y0, m.inv_g, m.H), cy1 = cell_coord(g.hiy, m.y0, m.inv_g, m.H);
  const int cx0 = cell_coord(g.lox, m.x0, m.inv_g, m.W), cx1 = cell_coord(g.hix, m.x0, m.inv_g, m.W);
  unsigned long long best = ~0ull;
  for (int cy = cy0 + row0; cy <= cy1; cy += row_stride) {
    const int s = m.cell_start[cy * m.W + cx0];
    const int e = m.cell_start[cy * m.W + cx1 + 1];
    for (int i = s + tid; i < e; i += nthreads) {
      const float x = m.x[i], y = m.y[i];
      const int pm = m.perm[i];
      if (x >= g.lox && x < g.hix && y >= g.loy && y < g.hiy && pm > g.perm_gt) {
        const unsigned long long k = ((unsigned long long)(unsigned)pm << 32) | (unsigned)i;
        best = k < best ? k : best;
      }
    }
  }
#pragma unroll
  for (int msk = 32; msk >= 1; msk >>= 1) {
    const unsigned long long o = __shfl_xor(best, msk);
    best = o < best ? o : best;
  }
  return best;  // the wave's minimum in every lane
}

// The decision at the subtree root the scan found (map_first_of_two's loop body): which of the tied
// points A, B the nearest-neighbour search for q visits first is settled at their lowest common
// ancestor; otherwise the region shrinks to the side both lie on.  One thread.
__device__ void region_step(const MapView &m, MapTieWalk *st, unsigned long long key) {
  st->steps++;
  if (key == ~0ull) {  // empty region: cannot happen (A and B are inside); reported as unresolved
    st->done = 2;
    return;
  }
  const size_t sidx = (size_t)(key & 0xFFFFFFFFull);
  const int cperm = (int)(key >> 32);
  const float cx = m.x[sidx], cy = m.y[sidx];
  const int axis = st->axis;
  const float split = axis ? cy : cx;
  const float q = axis ? st->qy : st->qx;
  const bool near_is_left = (q - split) <= 0;
  const float ca = axis ? st->ay : st->ax, cb = axis ? st->by : st->bx;
  if (cperm == st->aperm || cperm == st->bperm) {
    // the other point lies in cur's subtree: it is visited before cur iff it is on the nearer side
    const bool cur_is_a = cperm == st->aperm;
    const bool other_left = (cur_is_a ? cb : ca) < split;
    const bool other_first = other_left == near_is_left;
    st->first = cur_is_a ? (other_first ? 1 : 0) : (other_first ? 0 : 1);
    st->done = 1;
    return;
  }
  const bool a_left = ca < split, b_left = cb < split;
  if (a_left != b_left) {
    st->first = (a_left == near_is_left) ? 0 : 1;
    st->done = 1;
    return;
  }
  if (a_left)
    st->hi[axis] = split;
  else
    st->lo[axis] = split;
  st->cur_perm = cperm;
  st->axis = axis ^ 1;
}

// The first M points of the cloud (original index < M) by original index: the top of the reference's
// insertion-built map tree consists of exactly these (a node's ancestors were all inserted before it),
// so the host can walk that part without the device (map_first_of_two, trg_engine.cpp).
__global__ __launch_bounds__(256) void k_collect_first(MapView m, int M, float *out_xy) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < m.n; i += (size_t)gridDim.x * blockDim.x) {
    const int pm = m.perm[i];
    if (pm < M) {
      out_xy[2 * pm] = m.x[i];
      out_xy[2 * pm + 1] = m.y[i];
    }
  }
}

// One step with the whole grid scanning (the top of the tree: regions of millions of points); the
// workgroup that finishes last takes the decision.
__global__ __launch_bounds__(256) void k_region_walk_grid(MapView m, MapTieWalk *st) {
  if (st->done) return;  // (every workgroup reads the same value: only the last one of a launch writes it)
  const WalkRegion g{st->lo[0], st->hi[0], st->lo[1], st->hi[1], st->cur_perm};
  __shared__ unsigned long long red[4];
  __shared__ int last;
  const unsigned long long wbest = region_scan_rows(m, g, (int)blockIdx.x, (int)gridDim.x, (int)threadIdx.x, 256);
  if (lane_id() == 0) red[threadIdx.x >> 6] = wbest;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long best = red[0];
    for (int i = 1; i < 4; ++i) best = red[i] < best ? red[i] : best;
    if (best != ~0ull) atomicMin(&st->key, best);
    __threadfence();
    last = atomicAdd(&st->ticket, 1) == (int)gridDim.x - 1;
  }
  __syncthreads();
  if (last && threadIdx.x == 0) {
    __threadfence();
    const unsigned long long key = __hip_atomic_load(&st->key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    st->key = ~0ull;
    st->ticket = 0;
    region_step(m, st, key);
  }
}

// The rest of the walk in one workgroup: after a dozen halvings the region holds a few thousand
// points, and a step is a scan of a few cell rows.
__global__ __launch_bounds__(1024) void k_region_walk_block(MapView m, MapTieWalk *st, int max_steps) {
  __shared__ unsigned long long red[16];
  __shared__ int done_s;
  for (int it = 0; it < max_steps; ++it) {
    if (threadIdx.x == 0) done_s = st->done;
    __syncthreads();
    if (done_s) return;
    const WalkRegion g{st->lo[0], st->hi[0], st->lo[1], st->hi[1], st->cur_perm};
    // a wave per cell row, its lanes along the row
    const unsigned long long wbest = region_scan_rows(m, g, (int)(threadIdx.x >> 6), 16, (int)lane_id(), WAVE);
    if (lane_id() == 0) red[threadIdx.x >> 6] = wbest;
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned long long best = red[0];
      for (int i = 1; i < 16; ++i) best = red[i] < best ? red[i] : best;
      region_step(m, st, best);
      __threadfence_block();
    }
    __syncthreads();
  }
}

// LDS per wave of the edge kernels: x, y, z tile + hit buffer (4 * TCAP floats = 8 KB); the hit
// buffer doubles as the scratch of the global-memory fallback, whose capacity is therefore TCAP.
static_assert(TCAP % WAVE == 0, "tile capacity is a whole number of wave sweeps");
constexpr int EDGE_WAVES_PER_SIMD = 6;  // register budget of the edge kernels (<= 80 VGPRs); 5 and 8 measured slower
struct EdgeLds {
  WaveTile w[QW];
};

__global__ __launch_bounds__(QW *WAVE, EDGE_WAVES_PER_SIMD) void k_edges(MapView m, QueryParams p, const float *p1,
                                                    const float *p2, int count, float *mid,
                                                    DeviceCounters *ctr) {
  __shared__ EdgeLds lds;
  const int w = threadIdx.x >> 6;
  const int q = blockIdx.x * QW + w;
  if (q >= count) return;
  float *rec = mid + (size_t)q * MID_STRIDE;
  const EdgeMidOut o = edge_gather(m, p, p1[3 * q], p1[3 * q + 1], p1[3 * q + 2], p2[3 * q],
                                   p2[3 * q + 1], p2[3 * q + 2], lds.w[w], rec, ctr);
  store_mid(rec, o);
}

__global__ __launch_bounds__(QW *WAVE, EDGE_WAVES_PER_SIMD) void k_spec_edges(MapView m, QueryParams p,
                                                         const float *node_xyz, int count,
                                                         const int *n_acc, const float *sx,
                                                         const float *sy, const float *sz,
                                                         float *mid, DeviceCounters *ctr) {
  __shared__ EdgeLds lds;
  const int w = threadIdx.x >> 6;
  const int slot = blockIdx.x * QW + w;
  const int S = p.sample_num;
  if (slot >= count * S) return;
  const int node = slot / S;
  const int j = slot - node * S;
  if (j >= n_acc[node]) return;
  float *rec = mid + (size_t)slot * MID_STRIDE;
  const EdgeMidOut o = edge_gather(m, p, node_xyz[3 * node], node_xyz[3 * node + 1],
                                   node_xyz[3 * node + 2], sx[slot], sy[slot], sz[slot], lds.w[w],
                                   rec, ctr);
  store_mid(rec, o);
}

// Phase 2: one thread per edge: covariance -> SVD -> risk weight (trg.cpp:339-363).
// valid_n_acc != nullptr: slot layout of a chunk (node * S + j), slots with j >= n_acc are skipped.
__global__ __launch_bounds__(256) void k_edge_finish(const float *mid, int count, int S,
                                                     const int *valid_n_acc, int *status,
                                                     int *n_pts, float *weight, float *dist,
                                                     DeviceCounters *ctr, int which_counter) {
  __shared__ int wave_hits[4];
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  bool live = q < count;
  if (live && valid_n_acc) {
    const int node = q / S;
    live = (q - node * S) < valid_n_acc[node];
  }
  const float *r = mid + (size_t)(live ? q : 0) * MID_STRIDE;
  // one atomic per block for the instrumentation counter
  const int h = wave_sum(live ? __float_as_int(r[MID_HITS]) : 0);
  if (lane_id() == 0) wave_hits[threadIdx.x >> 6] = h;
  __syncthreads();
  if (threadIdx.x == 0 && ctr) {
    const unsigned long long tot =
        (unsigned long long)(wave_hits[0] + wave_hits[1] + wave_hits[2] + wave_hits[3]);
    DeviceCounters *c = &ctr[blockIdx.x % COUNTER_SHARDS];
    atomicAdd(which_counter ? &c->spec_hits : &c->edge_hits, tot);
  }
  if (!live) return;
  int st = __float_as_int(r[0]);
  float w = 0.0f;
  if ((st & EDGE_STATUS_MASK) == EDGE_OK) {
    float cov[3][3];
    mid_covariance(r, cov);
    int clamped = 0;
    w = risk_weight(cov, clamped);
    st |= clamped;
  }
  status[q] = st;
  if (n_pts) n_pts[q] = __float_as_int(r[1]);
  weight[q] = w;
  dist[q] = r[2];
}

#ifndef TRG_SW
#define TRG_SW 8
#endif
constexpr int SW = TRG_SW;  // waves per block in the sampling kernel = trials evaluated per round

// Block-shared neighbour tile of the sampling kernel.  Every draw of a node lies on the circle of
// radius expand_dist around it, so all its collision discs fall inside one (2*(d+r))^2 box: the
// block stages the box's points (x, y, z, original index) and the box's cell offsets ONCE, and
// each wave then reads only the few cell-row segments of its own disc out of LDS.
constexpr int STILE = 768;   // points in the block tile
constexpr int SBOX = 10;     // box rows / columns of cells the tile can describe
constexpr int SHCAP = 256;   // hits per disc kept for the median (denser discs: global fallback)

struct SampleLds {
  f2 xy[STILE];  // (x, y) interleaved: one 8-byte LDS read feeds the distance arithmetic
  float z[STILE];
  int perm[STILE];
  int cs[SBOX][SBOX + 1];  // cs[row][col]: tile offset of the first point of box cell (row, col)
  float zb[SW][SHCAP];
  int bx0, by0, ncols, nrows, total;
  int s_row[SBOX], row_off[SBOX + 1], row_len[SBOX];  // staging scratch
};

// isCollision + nearest map point of one disc, candidates read from the block tile
__device__ Disc sample_tile_disc(const MapView &m, const SampleLds &L, float *zb, float qx, float qy,
                                 float r, float h) {
  const int lane = lane_id();
  const float r2 = r * r;
  const CellRange c = cells_for(m, qx, qy, r);
  // the disc's cells in box coordinates (the box covers them by construction; clamp for safety)
  const int cx0 = max(c.cx0 - L.bx0, 0), cx1 = min(c.cx1 - L.bx0, L.ncols - 1);
  const int cy0 = max(c.cy0 - L.by0, 0), cy1 = min(c.cy1 - L.by0, L.nrows - 1);
  int n = 0;
  float best_d2 = FLT_MAX, best_z = 0.0f;
  int best_perm = INT_MAX;
  bool lane_tie = false;
  for (int row = cy0; row <= cy1; ++row) {
    const int s = L.cs[row][cx0], e = L.cs[row][cx1 + 1];
    for (int base = s; base < e; base += WAVE) {
      const int i = base + lane;
      bool hit = false;
      float z = 0.0f, d2 = 0.0f;
      if (i < e) {
        const f2 pt = L.xy[i];
        const float dx = pt.x - qx;
        const float dy = pt.y - qy;
        z = L.z[i];
        d2 = dx * dx + dy * dy;
        hit = d2 <= r2;
      }
      const unsigned long long mask = ballot(hit);
      if (hit) {
        const int pos = n + __popcll(mask & lanemask_lt());
        if (pos < SHCAP) zb[pos] = z;
        const int pm = L.perm[i];
        if (d2 == best_d2) lane_tie = true;
        if (d2 < best_d2) lane_tie = false;
        if (d2 < best_d2 || (d2 == best_d2 && pm < best_perm)) {
          best_d2 = d2;
          best_z = z;
          best_perm = pm;
        }
      }
      n += __popcll(mask);
    }
  }
  Disc out;
  out.n = n;
  out.cnt = 0;
  out.nn_z = 0.0f;
  out.nn_tie = 0;
  out.nn_d2 = 0.0f;
  if (n == 0) return out;
  // nearest point: smallest distance (d2 >= 0, so its bit pattern orders like the value), then the
  // smallest original index among the lanes that hold it -- DPP reductions, no LDS round trips
  const unsigned dkey = __float_as_uint(best_d2);
  const unsigned wdk = wave_reduce_key<true>(dkey);
  const bool holds = best_perm != INT_MAX && dkey == wdk;
  const unsigned wp = wave_reduce_key<true>(holds ? (unsigned)best_perm : 0xFFFFFFFFu);
  const unsigned long long who = ballot(holds && (unsigned)best_perm == wp);
  out.nn_z = __shfl(best_z, __ffsll((long long)who) - 1);
  out.nn_d2 = __uint_as_float(wdk);
  out.nn_tie = ballot(holds && ((unsigned)best_perm != wp || lane_tie)) != 0ull;
  if (n > SHCAP) {
    out.cnt = -1;  // caller falls back to the global-memory query
    return out;
  }
  // z range of the hits, from the compacted hit buffer (one value per lane in the usual case)
  wave_lds_sync();
  const float zl = zb[lane < n ? lane : 0];
  unsigned kmn = 0xFFFFFFFFu, kmx = 0u;
  for (int i = lane; i < n; i += WAVE) {
    const unsigned zk = float_key(i == lane ? zl : zb[i]);
    kmn = zk < kmn ? zk : kmn;
    kmx = zk > kmx ? zk : kmx;
  }
  const unsigned klo = wave_reduce_key<true>(kmn), khi = wave_reduce_key<false>(kmx);
  // every |z - z_med| <= zmax - zmin (rounding is monotone), so a flat disc needs no median
  if (!(key_float(khi) - key_float(klo) <= h))
    out.cnt = (n <= WAVE) ? median_count_lanes(zl, n, h, klo, khi) : median_count(zb, n, h);
  wave_lds_sync();
  return out;
}

// Stage the box of all possible trial discs of one queued node into the block tile (all threads of
// the workgroup call this; contains barriers).  Returns false when the box does not fit: the caller
// then uses the global-memory queries.
__device__ bool stage_sample_tile(const MapView &m, const QueryParams &p, float px, float py,
                                  SampleLds &L) {
  const int tid = threadIdx.x;
  const CellRange box = cells_for(m, px, py, p.expand_dist + p.robot_size * 1.002f + 1e-5f);
  const int ncols = box.cx1 - box.cx0 + 1, nrows = box.cy1 - box.cy0 + 1;
  bool use_tile = ncols <= SBOX && nrows <= SBOX;
  if (use_tile) {
    // absolute cell offsets of the box (one load per thread), then row extents
    const int ncs = nrows * (ncols + 1);
    if (tid < ncs) {
      const int row = tid / (ncols + 1), col = tid - row * (ncols + 1);
      L.cs[row][col] = m.cell_start[(box.cy0 + row) * m.W + box.cx0 + col];
    }
    __syncthreads();
    // Every trial disc lies within R = d + r of the node: of each cell row only the columns that
    // such a disc can reach are staged (the corners of the box are never read).  The bounds are
    // conservative by millimetres; which points a disc accepts is decided by the exact test.
    if (tid < nrows) {
      const int row = tid;
      const float R = (p.expand_dist + p.robot_size) * 1.002f + 2e-3f;
      const float g = 1.0f / m.inv_g;
      const float yc = m.y0 + ((float)(box.cy0 + row) + 0.5f) * g;
      const float dy = fmaxf(fabsf(yc - py) - 0.5f * g - 2e-3f, 0.0f);
      int c_lo = 0, c_hi = -1;  // empty row
      if (dy < R) {
        const float hx = sqrtf(R * R - dy * dy) + 2e-3f;
        c_lo = max(cell_coord(px - hx, m.x0, m.inv_g, m.W) - box.cx0, 0);
        c_hi = min(cell_coord(px + hx, m.x0, m.inv_g, m.W) - box.cx0, ncols - 1);
      }
      const int a = c_hi >= c_lo ? L.cs[row][c_lo] : L.cs[row][0];
      const int bnd = c_hi >= c_lo ? L.cs[row][c_hi + 1] : L.cs[row][0];
      L.s_row[row] = a;
      L.row_len[row] = bnd - a;
    }
    __syncthreads();
    if (tid == 0) {
      int acc = 0;
      for (int row = 0; row < nrows; ++row) {
        L.row_off[row] = acc;
        acc += L.row_len[row];
      }
      L.row_off[nrows] = acc;
      L.total = acc;
      L.bx0 = box.cx0;
      L.by0 = box.cy0;
      L.ncols = ncols;
      L.nrows = nrows;
    }
    __syncthreads();
    use_tile = L.total <= STILE;
  }
  if (use_tile) {
    const int total = L.total;
    for (int idx = tid; idx < total; idx += (int)blockDim.x) {
      int row = 0;
      for (int rr = 1; rr < nrows; ++rr) row += (idx >= L.row_off[rr]);
      const int src = L.s_row[row] + (idx - L.row_off[row]);
      const float4 r = m.pt[src];  // (x, y, z, original index) in one load
      f2 pt;
      pt.x = r.x;
      pt.y = r.y;
      L.xy[idx] = pt;
      L.z[idx] = r.z;
      L.perm[idx] = __float_as_int(r.w);
    }
    // cell offsets relative to the tile
    const int ncs = nrows * (ncols + 1);
    __syncthreads();
    if (tid < ncs) {
      const int row = tid / (ncols + 1), col = tid - row * (ncols + 1);
      // cells left / right of the staged columns are empty ranges at the row's start / end
      const int len = L.row_off[row + 1] - L.row_off[row];
      L.cs[row][col] = L.row_off[row] + min(max(L.cs[row][col] - L.s_row[row], 0), len);
    }
    __syncthreads();
  }

  return use_tile;
}

// One block per queued node: the rejection-sampling loop of expandGraph (trg.cpp:384-403).
// Each round evaluates SW consecutive draws concurrently (one wave per draw); acceptance is
// then decided in draw order, so the result equals the sequential loop's.
__global__ __launch_bounds__(SW *WAVE, 8) void k_sample_nodes(MapView m, QueryParams p,
                                                           const float *cos_t, const float *sin_t,
                                                           int table_bits, uint32_t seed,
                                                           uint32_t epoch, const float *node_xy,
                                                           const int *node_id, int count,
                                                           int *n_acc_out, int *n_draws_out,
                                                           float *sx, float *sy, float *sz,
                                                           DeviceCounters *ctr, const int *front,
                                                           const float *gnx, const float *gny,
                                                           int *mt_count, MapTieRec *mt_rec,
                                                           const int *count_dev, int node_base) {
  __shared__ SampleLds L;
  __shared__ int r_col[2][SW];
  __shared__ unsigned long long r_hits[SW];
  __shared__ int r_ties[SW];
  // count_dev: the frontier size lives on the device (launch issued before the host knew it; the
  // grid is an upper bound), node_base: first node of a follow-up launch
  const int node = (int)blockIdx.x + node_base;
  if (count_dev) count = min(count, *count_dev);
  if (node >= count) return;
  const int tid = threadIdx.x;
  const int w = tid >> 6;
  const int lane = lane_id();
  // two addressing modes: chunk arrays (host replay) or frontier ids into the device node arrays
  float px, py;
  uint32_t id;
  if (front) {
    const int g = front[node];
    px = gnx[g];
    py = gny[g];
    id = (uint32_t)g;
  } else {
    px = node_xy[2 * node];
    py = node_xy[2 * node + 1];
    id = (uint32_t)node_id[node];
  }

  const bool use_tile = stage_sample_tile(m, p, px, py, L);

  const int S = p.sample_num;
  const int max_trial_sample = 1000;
  int n_acc = 0, rejects = 0, draws = 0, round = 0;
  unsigned long long hits = 0;
  int ties = 0;
  const int wu = __builtin_amdgcn_readfirstlane(w);  // the wave index as a scalar
  // the table entries of a round are fetched one round ahead (a full round always consumes
  // exactly SW draws, so the next round's trial numbers are known)
  float c_next, s_next;
  {
    const uint32_t k = sample_hash(seed, epoch, id, (uint32_t)w) >> (32 - table_bits);
    c_next = cos_t[k];
    s_next = sin_t[k];
  }
  while (n_acc < S && rejects <= max_trial_sample) {
    const float qx = px + p.expand_dist * c_next;
    const float qy = py + p.expand_dist * s_next;
    {
      const uint32_t k = sample_hash(seed, epoch, id, (uint32_t)(draws + SW + w)) >> (32 - table_bits);
      c_next = cos_t[k];
      s_next = sin_t[k];
    }
    Disc d;
    bool done_tile = false;
    if (use_tile) {
      d = sample_tile_disc(m, L, L.zb[w], qx, qy, p.robot_size, p.height_threshold);
      done_tile = d.cnt >= 0;
    }
    // dense disc or oversized box: the general query (its hit buffer is this wave's LDS slice; a
    // disc with more hits than that selects the median by re-reading the map)
    if (!done_tile)
      d = disc_query<true>(m, qx, qy, p.robot_size, p.height_threshold, L.zb[w], ctr, SHCAP);
    const bool in_core = qx >= p.core_x0 && qx < p.core_x1 && qy >= p.core_y0 && qy < p.core_y1;
    const bool collides = disc_collides(d, p.collision_threshold) || !in_core;
    int *colbuf = r_col[round & 1];  // double-buffered: one barrier per round
    if (lane == 0) colbuf[w] = collides ? 1 : 0;
    __syncthreads();
    // In-order acceptance (trg.cpp:386-402).  The round's reject flags are packed into one mask so
    // that the sequential loop runs on the scalar unit; the wave whose draw is accepted stores it.
    const unsigned long long colmask = ballot(lane < SW && colbuf[lane < SW ? lane : 0] != 0);
    int my_slot = -1;
    int consumed = 0;  // draws of this round the sequential loop really made
    for (int i = 0; i < SW; ++i) {
      if (n_acc >= S || rejects > max_trial_sample) break;
      draws++;
      consumed++;
      if ((colmask >> i) & 1ull) {
        rejects++;
      } else {
        if (i == wu) my_slot = n_acc;
        n_acc++;
      }
    }
    // instrumentation counts the reference's queries only: a draw evaluated past the loop's
    // stopping point was never issued by expandGraph (trg.cpp:386-402)
    if (wu < consumed) hits += (unsigned long long)d.n;
    if (my_slot >= 0 && lane == 0) {
      const int slot = node * S + my_slot;
      sx[slot] = qx;
      sy[slot] = qy;
      sz[slot] = d.nn_z;
      if (d.nn_tie) {
        // the accepted sample's elevation hangs on a nearest-point tie: tell the host
        ties++;
        if (mt_count) {
          const int k = atomicAdd(mt_count, 1);
          if (k < MAPTIE_CAP) {
            MapTieRec rec;
            rec.slot = slot;
            rec.qx = qx;
            rec.qy = qy;
            rec.d2 = d.nn_d2;
            mt_rec[k] = rec;
          }
        }
      }
    }
    round++;
  }
  if (threadIdx.x == 0) {
    n_acc_out[node] = n_acc;
    n_draws_out[node] = draws;
  }
  if (lane == 0) {
    r_hits[w] = hits;
    r_ties[w] = ties;
  }
  __syncthreads();
  if (threadIdx.x == 0 && ctr) {
    unsigned long long th = 0;
    int tt = 0;
    for (int i = 0; i < SW; ++i) {
      th += r_hits[i];
      tt += r_ties[i];
    }
    DeviceCounters *c = &ctr[blockIdx.x % COUNTER_SHARDS];
    atomicAdd(&c->sample_hits, th);
    if (tt) atomicAdd(&c->nn_ties, (unsigned long long)tt);
  }
}

#include "trg_bfs.inc"
#include "trg_level.inc"
#include "trg_step3.inc"

inline int blocks_for(size_t n, int per_block, int cap) {
  size_t b = (n + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > (size_t)cap) b = cap;
  return (int)b;
}

}  // namespace

// ================================ launchers =====================================================

void launch_init_bounds(unsigned *d_bounds, hipStream_t s) {
  hipLaunchKernelGGL(k_init_bounds, dim3(1), dim3(64), 0, s, d_bounds);
}
void launch_bounds(const float *d_xyz, size_t n, size_t stride, unsigned *d_bounds, hipStream_t s) {
  hipLaunchKernelGGL(k_bounds, dim3(blocks_for(n, 256, 1024)), dim3(256), 0, s, d_xyz, n, stride,
                     d_bounds);
}
void launch_cell_count(const float *d_xyz, size_t n, size_t stride, float x0, float y0, float inv_g,
                       int W, int H, int *d_cell_of, int *d_rank, int *d_counts, hipStream_t s) {
  hipLaunchKernelGGL(k_cell_count, dim3(blocks_for(n, 256, 4096)), dim3(256), 0, s, d_xyz, n,
                     stride, x0, y0, inv_g, W, H, d_cell_of, d_rank, d_counts);
}
void launch_exclusive_scan(const int *d_counts, int *d_out, int m, int *d_tmp, hipStream_t s) {
  const int nt = (m + SCAN_TILE - 1) / SCAN_TILE;
  hipLaunchKernelGGL(k_scan_tiles, dim3(nt), dim3(256), 0, s, d_counts, d_out, m, d_tmp);
  hipLaunchKernelGGL(k_scan_tile_sums, dim3(1), dim3(256), 0, s, d_tmp, nt);
  hipLaunchKernelGGL(k_scan_add, dim3(nt), dim3(256), 0, s, d_out, m, d_tmp, nt);
}
void launch_scatter_sort_aos(const float *d_xyz, size_t n, size_t stride, const int *d_cell_of,
                             const int *d_rank, int ncell, const int *d_cell_start, void *d_aos, float *x,
                             float *y, float *z, int *perm, float4 *pt, hipStream_t s) {
  hipLaunchKernelGGL(k_scatter_aos, dim3(blocks_for(n, 256, 4096)), dim3(256), 0, s, d_xyz, n, stride,
                     d_cell_of, d_rank, d_cell_start, (float4 *)d_aos);
  hipLaunchKernelGGL(k_cell_sort_aos, dim3((ncell + 255) / 256), dim3(256), 0, s, ncell, d_cell_start,
                     (const float4 *)d_aos, x, y, z, perm, pt);
}

bool index_bins_plan(size_t n, size_t ncell, int *bin_shift, int *nbins, int *nwg) {
  int shift = 10;
  while (((ncell >> shift) + 1) > (size_t)BIN_MAX) ++shift;
  if ((1 << shift) > BIN_CELLS_MAX || n >= 0x7FFFFFF0u) return false;
  *bin_shift = shift;
  *nbins = (int)((ncell + ((size_t)1 << shift) - 1) >> shift);
  *nwg = (int)std::min<size_t>(BIN_WG, std::max<size_t>(1, (n + 8191) / 8192));
  return true;
}
void launch_index_bins(const float *d_xyz, size_t n, size_t stride, float x0, float y0, float inv_g, int W, int H,
                       int ncell, int bin_shift, int nbins, int nwg, int *d_hist, int *d_base, int *d_tmp,
                       float4 *d_scratch_a, float4 *d_scratch_b, int *d_cell_start, float *x, float *y, float *z,
                       int *perm, float4 *pt, hipStream_t s) {
  const size_t chunk = (n + nwg - 1) / nwg;
  hipLaunchKernelGGL(k_bin_count, dim3(nwg), dim3(BIN_THREADS), 0, s, d_xyz, n, stride, x0, y0, inv_g, W, H,
                     bin_shift, nbins, chunk, d_hist);
  launch_exclusive_scan(d_hist, d_base, nbins * nwg, d_tmp, s);
  hipLaunchKernelGGL(k_bin_scatter, dim3(nwg), dim3(BIN_THREADS), 0, s, d_xyz, n, stride, x0, y0, inv_g, W, H,
                     bin_shift, nbins, chunk, (const int *)d_base, d_scratch_a);
  hipLaunchKernelGGL(k_bin_cells, dim3(nbins), dim3(BIN_CELL_THREADS), 0, s, (const float4 *)d_scratch_a, (int)n,
                     x0, y0, inv_g, W, H, bin_shift, nbins, nwg, (const int *)d_base, ncell, d_cell_start,
                     d_scratch_b);
  hipLaunchKernelGGL(k_cell_sort_aos, dim3((ncell + 255) / 256), dim3(256), 0, s, ncell, (const int *)d_cell_start,
                     (const float4 *)d_scratch_b, x, y, z, perm, pt);
}

void launch_probe_collision(const MapView &m, QueryParams p, float threshold, const float *d_xy,
                            int count, int *flag, int *cnt, int *n, DeviceCounters *ctr,
                            hipStream_t s) {
  if (count <= 0) return;
  hipLaunchKernelGGL(k_probe_collision, dim3((count + QW - 1) / QW), dim3(QW * WAVE), 0, s, m, p,
                     threshold, d_xy, count, flag, cnt, n, ctr);
}
void launch_probe_nearest_z(const MapView &m, QueryParams p, const float *d_xy, int count, float *z,
                            int *found, DeviceCounters *ctr, hipStream_t s) {
  if (count <= 0) return;
  hipLaunchKernelGGL(k_probe_nearest_z, dim3((count + QW - 1) / QW), dim3(QW * WAVE), 0, s, m, p,
                     d_xy, count, z, found, ctr);
}
void launch_edges(const MapView &m, QueryParams p, const float *d_p1, const float *d_p2, int count,
                  float *mid, int *status, int *n_pts, float *weight, float *dist,
                  DeviceCounters *ctr, hipStream_t s) {
  if (count <= 0) return;
  hipLaunchKernelGGL(k_edges, dim3((count + QW - 1) / QW), dim3(QW * WAVE), 0, s, m, p, d_p1, d_p2,
                     count, mid, ctr);
  hipLaunchKernelGGL(k_edge_finish, dim3((count + 255) / 256), dim3(256), 0, s, mid, count, 1,
                     (const int *)nullptr, status, n_pts, weight, dist, ctr, 0);
}
void launch_sample_nodes(const MapView &m, QueryParams p, const float *cos_t, const float *sin_t,
                         int table_bits, uint32_t seed, uint32_t epoch, const float *node_xy,
                         const int *node_id, int count, int *n_acc, int *n_draws, float *sx,
                         float *sy, float *sz, DeviceCounters *ctr, int *mt_count,
                         MapTieRec *mt_rec, hipStream_t s) {
  if (count <= 0) return;
  hipLaunchKernelGGL(k_sample_nodes, dim3(count), dim3(SW * WAVE), 0, s, m, p, cos_t, sin_t,
                     table_bits, seed, epoch, node_xy, node_id, count, n_acc, n_draws, sx, sy, sz,
                     ctr, (const int *)nullptr, (const float *)nullptr, (const float *)nullptr,
                     mt_count, mt_rec, (const int *)nullptr, 0);
}
void launch_map_tied_set(const MapView &m, float qx, float qy, float r0, MapTieSet *d_out,
                         hipStream_t s) {
  hipLaunchKernelGGL(k_map_tied_set, dim3(1), dim3(WAVE), 0, s, m, qx, qy, r0, d_out);
}
void launch_collect_first(const MapView &m, int M, float *d_out_xy, hipStream_t s) {
  hipLaunchKernelGGL(k_collect_first, dim3(blocks_for(m.n, 256, 4096)), dim3(256), 0, s, m, M, d_out_xy);
}
void launch_map_tie_walk(const MapView &m, MapTieWalk *d_state, int grid_steps, int block_steps, hipStream_t s) {
  // the top of the tree with the whole grid (cell rows strided over the workgroups), the rest in one
  // workgroup
  const int rows = m.H < 256 ? m.H : 256;
  for (int k = 0; k < grid_steps; ++k)
    hipLaunchKernelGGL(k_region_walk_grid, dim3(rows), dim3(256), 0, s, m, d_state);
  if (block_steps > 0) hipLaunchKernelGGL(k_region_walk_block, dim3(1), dim3(1024), 0, s, m, d_state, block_steps);
}
void launch_spec_edges(const MapView &m, QueryParams p, const float *node_xyz, int count,
                       const int *n_acc, const float *sx, const float *sy, const float *sz,
                       float *mid, int *status, int *n_pts, float *weight, float *dist,
                       DeviceCounters *ctr, hipStream_t s) {
  if (count <= 0) return;
  const int slots = count * p.sample_num;
  hipLaunchKernelGGL(k_spec_edges, dim3((slots + QW - 1) / QW), dim3(QW * WAVE), 0, s, m, p,
                     node_xyz, count, n_acc, sx, sy, sz, mid, ctr);
  hipLaunchKernelGGL(k_edge_finish, dim3((slots + 255) / 256), dim3(256), 0, s, mid, slots,
                     p.sample_num, n_acc, status, n_pts, weight, dist, ctr, 1);
}

size_t edge_mid_floats(size_t edges) { return edges * MID_STRIDE; }

#include "trg_bfs_launch.inc"
#include "trg_stitch.inc"

}  // namespace trg
