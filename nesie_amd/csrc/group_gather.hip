// group_points / gather_points forward + backward for gfx950.
//
// Replaces group_points_kernel / group_points_grad_kernel
// (reference mmdet3d/ops/group_points/src/group_points_cuda.cu:56-80, :10-31) and
// gather_points_kernel / gather_points_grad_kernel
// (reference mmdet3d/ops/gather_points/src/gather_points_cuda.cu:8-26, :51-70).
// gather is group with nsample == 1, so one kernel pair serves both.
//
// Forward is pure data movement (HBM/L2 bound): a thread owns one output
// column e = (p, s), reads idx[e] ONCE and walks CH_PER_THREAD channel rows, so
// the index traffic is amortised over the channels and every store is a dense
// 256-byte row segment per wave.  Backward: when a scene's n accumulators for a
// few channel rows fit LDS (pool.hip: group_bwd_lds_kernel) the adds go to LDS and
// each row is written once; otherwise the same walk with float atomics to HBM
// (the reference does the latter; sum order is not deterministic either way).
#include "common.h"
#include <stdlib.h>

namespace nesie {

constexpr int GG_BLOCK = 256;
constexpr int GG_CH = 8;  // channels walked per thread

// points (B,C,N), idx (B,E) -> out (B,C,E)        E = npoints * nsample
__global__ __launch_bounds__(GG_BLOCK) void group_fwd_kernel(
    int c, int n, int e_total, long long gstride, const float *__restrict__ points,
    const int *__restrict__ idx, float *__restrict__ out) {
  const int e = blockIdx.x * GG_BLOCK + threadIdx.x;
  const int c0 = blockIdx.y * GG_CH;
  const int bi = blockIdx.z;
  if (e >= e_total) return;
  int src = idx[(size_t)bi * e_total + e];
  src = src < 0 ? 0 : (src >= n ? n - 1 : src);  // never fault on a bad index
  const float *p = points + ((size_t)bi * c + c0) * n + src;
  float *o = out + (size_t)bi * gstride + (size_t)c0 * e_total + e;  // gstride: scene pitch
  const int cend = c - c0 < GG_CH ? c - c0 : GG_CH;
  float v[GG_CH];
  if (cend == GG_CH) {   // whole channel block (uniform): the eight gathers issue back to back
#pragma unroll
    for (int i = 0; i < GG_CH; ++i) v[i] = p[(size_t)i * n];
#pragma unroll
    for (int i = 0; i < GG_CH; ++i) o[(size_t)i * e_total] = v[i];
    return;
  }
#pragma unroll
  for (int i = 0; i < GG_CH; ++i)
    if (i < cend) v[i] = p[(size_t)i * n];
#pragma unroll
  for (int i = 0; i < GG_CH; ++i)
    if (i < cend) o[(size_t)i * e_total] = v[i];
}

// grad_out (B,C,E), idx (B,E) -> grad_points (B,C,N) += ...
__global__ __launch_bounds__(GG_BLOCK) void group_bwd_kernel(
    int c, int n, int e_total, long long gstride, const float *__restrict__ grad_out,
    const int *__restrict__ idx, float *__restrict__ grad_points) {
  const int e = blockIdx.x * GG_BLOCK + threadIdx.x;
  const int c0 = blockIdx.y * GG_CH;
  const int bi = blockIdx.z;
  if (e >= e_total) return;
  int dst = idx[(size_t)bi * e_total + e];
  dst = dst < 0 ? 0 : (dst >= n ? n - 1 : dst);  // never fault on a bad index
  const float *g = grad_out + (size_t)bi * gstride + (size_t)c0 * e_total + e;
  float *gp = grad_points + ((size_t)bi * c + c0) * n + dst;
  const int cend = c - c0 < GG_CH ? c - c0 : GG_CH;
  float v[GG_CH];
#pragma unroll
  for (int i = 0; i < GG_CH; ++i)
    if (i < cend) v[i] = g[(size_t)i * e_total];
#pragma unroll
  for (int i = 0; i < GG_CH; ++i)
    if (i < cend) atomicAdd(gp + (size_t)i * n, v[i]);
}

int launch_group_bwd_lds(int b, int c, int n, long long e_total, long long gstride,
                         const float *grad_out, const int *idx, float *grad_points,
                         hipStream_t s);

// gstride = elements between scenes of the GROUPED tensor (out of the forward, grad_out of the
// backward); c * e_total for a dense one, larger when it is a channel slice of a wider tensor
static int launch_group(bool fwd, const char *W, int b, int c, int n, long long e_total,
                        const float *a, const int *idx, float *o, void *stream,
                        long long gstride = -1) {
  if (gstride < 0) gstride = (long long)c * e_total;
  NESIE_REQUIRE(b >= 0 && c >= 0 && n >= 0 && e_total >= 0, W);
  if (b == 0 || c == 0 || e_total == 0) return NESIE_OK;
  NESIE_REQUIRE(n >= 1 && a && idx && o, W);
  NESIE_REQUIRE(e_total < (1ll << 31) && b <= 65535 && cdiv(c, GG_CH) <= 65535, W);
  if (!fwd && n <= 16384 && e_total >= 4 * (long long)n)
    return launch_group_bwd_lds(b, c, n, e_total, gstride, a, idx, o, (hipStream_t)stream);
  dim3 grid(cdiv(e_total, GG_BLOCK), cdiv(c, GG_CH), b);
  if (fwd)
    hipLaunchKernelGGL(group_fwd_kernel, grid, dim3(GG_BLOCK), 0, (hipStream_t)stream, c,
                       n, (int)e_total, gstride, a, idx, o);
  else
    hipLaunchKernelGGL(group_bwd_kernel, grid, dim3(GG_BLOCK), 0, (hipStream_t)stream, c,
                       n, (int)e_total, gstride, a, idx, o);
  return check_launch(W);
}

// Scatter-add through an inverted index: order[b][.] lists the grouped columns sorted by their
// source point (each point's run in ascending column order) and src[b][.] the source point of
// each.  A LANE owns one sorted entry, a wave 64 consecutive ones; within the wave a segmented
// scan (keys are contiguous) sums the entries of each point.  EVERY RUN IS SUMMED BY EXACTLY ONE
// WAVE -- the one whose 64 entries hold the run's first entry: it follows a run that leaves its
// chunk through the next chunks (wave-uniform loop, fixed butterfly order), and the waves of those
// chunks skip the entries that continue a run from before.  So the result of a point is one plain
// read-modify-write by one lane: no float atomics, and the sum order depends on the index alone --
// two launches on the same inputs give the same bits (round 3 added a wave's share of a run with a
// float atomic: runs longer than 64 entries -- low-index points sit in a hundred balls, ball query
// keeps the FIRST nsample hits -- met three or more adds in arrival order).  Work per wave is the
// same whatever the run lengths, except for the owner of a long run.  The index depends on the
// coordinates only, so a training loop builds it ahead of the step together with the ball query.
__global__ __launch_bounds__(GG_BLOCK) void group_bwd_csr_kernel(
    int c, int n, int e_total, long long gstride, int ediv, const float *__restrict__ grad_out,
    const float *__restrict__ weight, const int *__restrict__ order,
    const int *__restrict__ src, float *__restrict__ grad_points) {
  // ediv / weight: the entries are (column, tap) pairs, ediv taps per column of grad_out, each
  // scaled by weight[entry] (three_interpolate's backward); ediv = 1, weight = NULL for groups
  const int e = blockIdx.x * GG_BLOCK + threadIdx.x;
  const int c0 = blockIdx.y * GG_CH;
  const int bi = blockIdx.z;
  const int lane = threadIdx.x & 63;
  const int base = e - lane;                               // first entry of this wave's chunk
  if (base >= e_total) return;                             // (wave-uniform)
  const int *sb = src + (size_t)bi * e_total;
  const int *ob = order + (size_t)bi * e_total;
  const bool live = e < e_total;
  const int ee = live ? e : e_total - 1;
  const int col = ob[ee];
  const int key = live ? sb[ee] : -1;
  const int prev = base > 0 ? sb[base - 1] : -2;           // (uniform) the entry in front of the chunk
  const bool cont = live && key == prev;                   // continues a run an earlier wave owns
  // same[d]: the entry d lanes below belongs to the same point (then so do all in between)
  bool same[6];
#pragma unroll
  for (int d = 0; d < 6; ++d) {
    const int kd = __shfl_up(key, 1 << d, 64);
    same[d] = lane >= (1 << d) && kd == key;
  }
  const int knext = __shfl_down(key, 1, 64);
  const bool tail = live && !cont && (lane == 63 || knext != key);
  const int ncols = e_total / ediv;
  const float *gbase = grad_out + (size_t)bi * gstride + (size_t)c0 * ncols;
  const float *wb = weight ? weight + (size_t)bi * e_total : nullptr;
  const float wgt = wb ? wb[col] : 1.f;
  const float *g = gbase + col / ediv;
  float *dst = grad_points + ((size_t)bi * c + c0) * n + (key < 0 ? 0 : key);
  const int cend = c - c0 < GG_CH ? c - c0 : GG_CH;
  float v[GG_CH];
#pragma unroll
  for (int i = 0; i < GG_CH; ++i)
    v[i] = (live && i < cend) ? __fmul_rn(g[(size_t)i * ncols], wgt) : 0.f;
#pragma unroll
  for (int i = 0; i < GG_CH; ++i) {
#pragma unroll
    for (int d = 0; d < 6; ++d) {
      const float t = __shfl_up(v[i], 1 << d, 64);
      if (same[d]) v[i] += t;
    }
  }
  // the chunk's LAST run may go on in the following chunks: its owner (this wave, unless the run
  // itself came from before) collects the rest, 64 entries per trip
  const int klast = __builtin_amdgcn_readlane(key, 63);
  const int clast = __builtin_amdgcn_readlane((int)cont, 63);
  if (klast >= 0 && !clast && base + 64 < e_total && sb[base + 64] == klast) {   // (uniform)
    float extra[GG_CH];
#pragma unroll
    for (int i = 0; i < GG_CH; ++i) extra[i] = 0.f;
    for (int pos = base + 64; pos < e_total; pos += 64) {
      const int e2 = pos + lane;
      const bool in = e2 < e_total && sb[e2 < e_total ? e2 : e_total - 1] == klast;
      const unsigned long long m = __ballot(in);
      if (m == 0ull) break;
      const int col2 = ob[in ? e2 : pos];
      const float w2 = wb ? wb[col2] : 1.f;
#pragma unroll
      for (int i = 0; i < GG_CH; ++i) {
        float t = (in && i < cend) ? __fmul_rn(gbase[(size_t)i * ncols + col2 / ediv], w2) : 0.f;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) t += __shfl_xor(t, off, 64);
        extra[i] += t;
      }
      if (m != ~0ull) break;                               // the run ended inside this trip
    }
#pragma unroll
    for (int i = 0; i < GG_CH; ++i) v[i] += lane == 63 ? extra[i] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < GG_CH; ++i)
    if (tail && i < cend) dst[(size_t)i * n] += v[i];      // one lane per point in the whole launch
}

// The same scatter with the gradient rows IN LDS (round 5).  group_bwd_csr_kernel gathers
// grad_out[channel][order[e]] from HBM: neighbouring lanes (sorted by SOURCE point) read columns
// that lie anywhere in the (M, ns) row, 4 bytes per 64-byte line -- SA2's 137 MB gradient moved
// ~2 GB and the launch took 135 us (six launches per step: 0.39 ms) -- and the wave that owns a
// long run (a low-index point sits in a thousand balls: ball query keeps the FIRST nsample hits)
// follows it alone through dozens of chunks, two dependent loads per trip.  Here a 1024-thread
// workgroup owns CH whole channel rows of one scene and streams them into LDS once (dense 16-byte
// loads); the run bounds of every point come from one pass over the sorted source list (LDS); then
//   * a THREAD owns each point with a run of <= GR_LONG entries and adds its gathered values in
//     ascending entry order -- the order of the reference's serial loop restated in the oracle
//     (SURVEY appendix A.3): five instructions per entry where the segmented scan spent forty;
//   * a WAVE owns each longer run: lane l adds entries l, l + 64, ... in ascending order, a fixed
//     butterfly adds the 64 partials.
// Every point is written once (a point without entries gets its zero here: no fill needed), in an
// order fixed by the index alone: bitwise reproducible.
constexpr int GR_BLOCK = 1024, GR_LPP = 4, GR_LONG = 64 * GR_LPP;

// EDIV = taps per gradient column (1: grouping, 3: three_interpolate's weighted taps), compile-time:
// a runtime division per entry was a third of the loop
template <int CH, int EDIV>
__global__ __launch_bounds__(GR_BLOCK) void group_bwd_csr_rows_kernel(
    int c, int n, int e_total, long long gstride, int ediv, const float *__restrict__ grad_out,
    const float *__restrict__ weight, const int *__restrict__ order,
    const int *__restrict__ src, float *__restrict__ grad_points) {
  extern __shared__ __attribute__((aligned(16))) float gr_lds[];
  const int c0 = blockIdx.x * CH, bi = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ncols = e_total / EDIV;
  const int ncols4 = (ncols + 3) & ~3;
  float *rows = gr_lds;                                   // [CH][ncols4]
  int *starts = (int *)(rows + (size_t)CH * ncols4);      // [n]
  int *ends = starts + n;                                 // [n]
  int *longs = ends + n;                                  // [n]: points with long runs; [n] = their count
  const int cend = c - c0 < CH ? c - c0 : CH;
  const float *gbase = grad_out + (size_t)bi * gstride + (size_t)c0 * ncols;
  const int *sb = src + (size_t)bi * e_total;
  const int *ob = order + (size_t)bi * e_total;
  const float *wb = weight ? weight + (size_t)bi * e_total : nullptr;
  for (int d = tid; d < n; d += GR_BLOCK) starts[d] = ends[d] = 0;
  if (tid == 0) longs[n] = 0;
  if ((((uintptr_t)gbase) & 15) == 0 && (ncols & 3) == 0) {     // rows are contiguous in the gradient
    const float4 *g4 = (const float4 *)gbase;
    float4 *r4 = (float4 *)rows;
    const int total4 = cend * ncols / 4;
    int i = tid;
    for (; i + 7 * GR_BLOCK < total4; i += 8 * GR_BLOCK) {      // eight 16-byte loads in flight per thread
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = g4[i + u * GR_BLOCK];
#pragma unroll
      for (int u = 0; u < 8; ++u) r4[i + u * GR_BLOCK] = v[u];
    }
    for (; i < total4; i += GR_BLOCK) r4[i] = g4[i];
  } else {
    for (int i = tid; i < cend * ncols; i += GR_BLOCK) rows[(i / ncols) * ncols4 + i % ncols] = gbase[i];
  }
  __syncthreads();
  {   // run bounds from the sorted source list: a thread owns four consecutive entries
    for (int e0 = 4 * tid; e0 < e_total; e0 += 4 * GR_BLOCK) {
      int sv[6];
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        const int e = e0 - 1 + u;
        sv[u] = (e >= 0 && e < e_total) ? sb[e] : -1;
      }
#pragma unroll
      for (int u = 1; u < 5; ++u) {
        const int e = e0 - 1 + u;
        if (e < e_total && sv[u] >= 0 && sv[u] < n) {
          if (sv[u] != sv[u - 1]) starts[sv[u]] = e;
          if (sv[u] != sv[u + 1]) ends[sv[u]] = e + 1;
        }
      }
    }
  }
  __syncthreads();
  float *dst0 = grad_points + ((size_t)bi * c + c0) * n;
  // GR_LPP lanes per point: lane l adds entries st + l, st + l + LPP, ... in ascending order (four index
  // loads in flight), the LPP partials meet in a fixed butterfly
  const int sub = tid % GR_LPP;
  for (int d0 = 0; d0 < n; d0 += GR_BLOCK / GR_LPP) {
    const int d = d0 + tid / GR_LPP;
    const bool have = d < n;
    const int st = have ? starts[d] : 0, en = have ? ends[d] : 0;
    const bool is_long = en - st > GR_LONG;
    if (is_long && sub == 0) longs[atomicAdd(&longs[n], 1)] = d;   // (any order: every long run has one owner wave)
    float acc[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) acc[i] = 0.f;
    if (!is_long) {
      int j = st + sub;
      for (; j + 3 * GR_LPP < en; j += 4 * GR_LPP) {
        int col[4];
        float w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) col[u] = ob[j + u * GR_LPP];
#pragma unroll
        for (int u = 0; u < 4; ++u) w[u] = wb ? wb[col[u]] : 1.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float *g = rows + col[u] / EDIV;
#pragma unroll
          for (int i = 0; i < CH; ++i) acc[i] += __fmul_rn(g[i * ncols4], w[u]);
        }
      }
      for (; j < en; j += GR_LPP) {
        const int col = ob[j];
        const float w = wb ? wb[col] : 1.f;
        const float *g = rows + col / EDIV;
#pragma unroll
        for (int i = 0; i < CH; ++i) acc[i] += __fmul_rn(g[i * ncols4], w);
      }
    }
#pragma unroll
    for (int i = 0; i < CH; ++i) {
#pragma unroll
      for (int off = GR_LPP / 2; off >= 1; off >>= 1) acc[i] += __shfl_xor(acc[i], off, 64);
    }
    if (have && !is_long && sub == 0) {
#pragma unroll
      for (int i = 0; i < CH; ++i)
        if (i < cend) dst0[(size_t)i * n + d] = acc[i];
    }
  }
  __syncthreads();
  const int nlong = longs[n];
  for (int li = wave; li < nlong; li += GR_BLOCK / 64) {
    const int d = longs[li];
    const int st = starts[d], en = ends[d];
    float acc[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) acc[i] = 0.f;
    for (int j = st + lane; j < en; j += 64) {
      const int col = ob[j];
      const float w = wb ? wb[col] : 1.f;
      const float *g = rows + col / EDIV;
#pragma unroll
      for (int i = 0; i < CH; ++i) acc[i] += __fmul_rn(g[i * ncols4], w);
    }
#pragma unroll
    for (int i = 0; i < CH; ++i) {
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) acc[i] += __shfl_xor(acc[i], off, 64);
    }
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < CH; ++i)
        if (i < cend) dst0[(size_t)i * n + d] = acc[i];
    }
  }
}

__global__ __launch_bounds__(256) void zero_fill_kernel(float *__restrict__ p, long long n) {
  const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i + 3 < n) *(float4 *)(p + i) = make_float4(0.f, 0.f, 0.f, 0.f);
  else for (long long j = i; j < n; ++j) p[j] = 0.f;
}
// (a kernel, not hipMemsetAsync: a memset node beside a kernel that adds into the same buffer lost its
// ordering on later replays of a captured graph on this runtime)
static void zero_fill(float *p, long long n, hipStream_t s) {
  if (n > 0) hipLaunchKernelGGL(zero_fill_kernel, dim3(cdiv(cdiv(n, 4), 256)), dim3(256), 0, s, p, n);
}

// -> launched (true) or not applicable (false: the caller uses group_bwd_csr_kernel)
static bool launch_group_bwd_csr_rows(int b, int c, int n, long long e_total, long long gstride, int ediv,
                                      const float *grad_out, const float *weight, const int *order,
                                      const int *src, float *grad_points, hipStream_t s) {
  const char *sw = getenv("NESIE_CSR_ROWS");           // A/B switch, read per call (tests flip it)
  const int on = sw ? atoi(sw) : 1;
  const long long ncols = e_total / ediv;
  constexpr long long LDS_MAX = 156 * 1024;
  auto need = [&](int ch) { return ((long long)ch * ((ncols + 3) & ~3ll) + 3ll * n + 1) * 4; };
  if (!on || need(1) > LDS_MAX || b > 65535 || e_total % ediv || (ediv != 1 && ediv != 3)) return false;
  // rows per workgroup: as many as fit, at most 8, and no more than leaves >= 256 workgroups
  int ch = 8;
  while (ch > 1 && (need(ch) > LDS_MAX || (long long)cdiv(c, ch) * b < 256)) ch >>= 1;
  const size_t lds = (size_t)need(ch);
  const dim3 grid(cdiv(c, ch), b);
#define GRL(CH, ED)                                                                                     \
  do {                                                                                                  \
    static bool attr = false;                                                                           \
    if (!attr) {                                                                                        \
      (void)hipFuncSetAttribute((const void *)group_bwd_csr_rows_kernel<CH, ED>,                        \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX);              \
      attr = true;                                                                                      \
    }                                                                                                   \
    hipLaunchKernelGGL((group_bwd_csr_rows_kernel<CH, ED>), grid, dim3(GR_BLOCK), lds, s, c, n, (int)e_total, \
                       gstride, ediv, grad_out, weight, order, src, grad_points);                       \
  } while (0)
  if (ediv == 1) { if (ch == 8) GRL(8, 1); else if (ch == 4) GRL(4, 1); else if (ch == 2) GRL(2, 1); else GRL(1, 1); }
  else { if (ch == 8) GRL(8, 3); else if (ch == 4) GRL(4, 3); else if (ch == 2) GRL(2, 3); else GRL(1, 3); }
#undef GRL
  return true;
}

// Inverted index of idx[b][0..e_total) over n source points, one workgroup per scene: LDS
// histogram (integer ds_add), exclusive scan, then each column takes the next free slot of its
// point's run (returning ds_add).  Built on the side stream with the ball query: 8 workgroups,
// tens of microseconds.  The slots are claimed in arrival order into `scratch`; a second pass
// ranks every entry inside its run (runs are short: M ns / N on average), so `order` lists each
// run in ascending column order and the segmented sums of the backward are reproducible.
constexpr int II_BLOCK = 1024;
constexpr int II_WINDOW = 8192;     // source points per workgroup (LDS histogram + cursor)

// Any n (round 5): workgroup (scene, window w) owns the source points [w * II_WINDOW, + II_WINDOW):
// it first counts the entries that belong to EARLIER windows (that many positions of order / srcs lie
// in front of its own), then sorts its own entries as before.  Every window reads the whole index
// row, so the cost grows with the window count -- 5 windows at 40 000 points; the step itself only
// ever inverts indices over <= 2 048 points (the 40 000-point level carries no feature gradient).
__global__ __launch_bounds__(II_BLOCK) void inverted_index_kernel(
    int n, int e_total, const int *__restrict__ idx, int *__restrict__ order,
    int *__restrict__ srcs, int *__restrict__ scratch_out) {
  extern __shared__ int lds[];  // cursor[bins]; scan scratch [II_BLOCK]; base counter
  const int lo_pt = blockIdx.y * II_WINDOW;
  const int bins = n - lo_pt < II_WINDOW ? n - lo_pt : II_WINDOW;
  int *cur = lds, *scratch = lds + bins, *before = scratch + II_BLOCK;
  const int bi = blockIdx.x, tid = threadIdx.x;
  const int *ix = idx + (size_t)bi * e_total;
  for (int j = tid; j < bins; j += II_BLOCK) cur[j] = 0;
  if (tid == 0) *before = 0;
  __syncthreads();
  int mine_before = 0;
  for (int e = tid; e < e_total; e += II_BLOCK) {
    int s = ix[e];
    s = s < 0 ? 0 : (s >= n ? n - 1 : s);
    if (s < lo_pt) ++mine_before;
    else if (s < lo_pt + bins) atomicAdd(&cur[s - lo_pt], 1);
  }
  if (lo_pt > 0) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mine_before += __shfl_xor(mine_before, off, 64);
    if ((tid & 63) == 0 && mine_before) atomicAdd(before, mine_before);
  }
  __syncthreads();
  const int base = *before;
  // exclusive scan over the bins: each thread owns a contiguous chunk
  const int per = (bins + II_BLOCK - 1) / II_BLOCK;
  const int lo = tid * per, hi = lo + per < bins ? lo + per : bins;
  int sum = 0;
  for (int j = lo; j < hi; ++j) sum += cur[j];
  scratch[tid] = sum;
  __syncthreads();
  for (int d = 1; d < II_BLOCK; d <<= 1) {
    const int v = tid >= d ? scratch[tid - d] : 0;
    __syncthreads();
    scratch[tid] += v;
    __syncthreads();
  }
  int run = base + scratch[tid] - sum;  // first position of this chunk
  for (int j = lo; j < hi; ++j) {
    const int cnt = cur[j];
    cur[j] = run;  // becomes the cursor of point lo_pt + j
    run += cnt;
  }
  __syncthreads();
  int *ord = order + (size_t)bi * e_total;
  int *sr = srcs + (size_t)bi * e_total;
  int *tmp = scratch_out + (size_t)bi * e_total;
  for (int e = tid; e < e_total; e += II_BLOCK) {
    int s = ix[e];
    s = s < 0 ? 0 : (s >= n ? n - 1 : s);
    if (s >= lo_pt && s < lo_pt + bins) tmp[atomicAdd(&cur[s - lo_pt], 1)] = e;
  }
  __syncthreads();   // cur[j] is now the END of run j; run j starts where run j - 1 ends (run 0: at base)
  const int total = cur[bins - 1];
  for (int i = base + tid; i < total; i += II_BLOCK) {
    const int e = tmp[i];
    int s = ix[e];
    s = (s < 0 ? 0 : (s >= n ? n - 1 : s)) - lo_pt;
    sr[i] = s + lo_pt;
    const int lo_s = s > 0 ? cur[s - 1] : base, hi_s = cur[s];
    int rank = 0;
    for (int k = lo_s; k < hi_s; ++k) rank += tmp[k] < e ? 1 : 0;
    ord[lo_s + rank] = e;
  }
}

// The same index by a STABLE counting sort (n <= II_STABLE_N bins): no ranking pass.  Wave w of the
// 16 owns a contiguous range of entries and keeps its own histogram row cnt[w][.] in LDS; after the
// scan cnt[w][s] is where wave w's first entry of bin s goes (bin-major, wave-minor), and the wave
// places its entries 64 at a time in ascending order: the lanes that hold the same bin find each
// other with a ballot per distinct bin of the round and take consecutive positions in lane order.
// The result is `order` ascending inside every run by construction -- what the ranking pass of
// inverted_index_kernel computes in time quadratic in the run length.
constexpr int II_STABLE_N = 2048, II_WAVES = II_BLOCK / 64;

__global__ __launch_bounds__(II_BLOCK) void inverted_index_stable_kernel(
    int n, int e_total, const int *__restrict__ idx, int *__restrict__ order,
    int *__restrict__ srcs) {
  extern __shared__ int lds[];                       // cnt[II_WAVES][n], then scan scratch [II_BLOCK]
  int *cnt = lds, *scratch = lds + II_WAVES * n;
  const int bi = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int *ix = idx + (size_t)bi * e_total;
  for (int j = tid; j < II_WAVES * n; j += II_BLOCK) cnt[j] = 0;
  __syncthreads();
  const int per = ((e_total + II_WAVES - 1) / II_WAVES + 63) / 64 * 64;   // entries per wave, whole rounds
  const int e0 = wave * per, e1 = e0 + per < e_total ? e0 + per : e_total;
  int *mine = cnt + wave * n;
  for (int e = e0 + lane; e < e1; e += 64) {
    int s = ix[e];
    s = s < 0 ? 0 : (s >= n ? n - 1 : s);
    atomicAdd(&mine[s], 1);
  }
  __syncthreads();
  // totals per bin -> exclusive scan over the bins -> per-(wave, bin) start positions
  const int bper = (n + II_BLOCK - 1) / II_BLOCK;    // bins per thread (contiguous)
  const int b0 = tid * bper, b1 = b0 + bper < n ? b0 + bper : n;
  int sum = 0;
  for (int sbin = b0; sbin < b1; ++sbin)
    for (int w = 0; w < II_WAVES; ++w) sum += cnt[w * n + sbin];
  scratch[tid] = sum;
  __syncthreads();
  for (int d = 1; d < II_BLOCK; d <<= 1) {
    const int v = tid >= d ? scratch[tid - d] : 0;
    __syncthreads();
    scratch[tid] += v;
    __syncthreads();
  }
  int run = scratch[tid] - sum;                      // first position of this thread's first bin
  for (int sbin = b0; sbin < b1; ++sbin)
    for (int w = 0; w < II_WAVES; ++w) {
      const int c = cnt[w * n + sbin];
      cnt[w * n + sbin] = run;
      run += c;
    }
  __syncthreads();
  int *ord = order + (size_t)bi * e_total;
  int *sr = srcs + (size_t)bi * e_total;
  volatile int *cur = mine;                          // this wave's cursors (only this wave touches them)
  for (int base = e0; base < e1; base += 64) {
    const int e = base + lane;
    const bool live = e < e1;
    int s = live ? ix[e] : 0;
    s = s < 0 ? 0 : (s >= n ? n - 1 : s);
    unsigned long long todo = __ballot(live);
    while (todo) {                                   // one trip per distinct bin of the round
      const int leader = __builtin_ctzll(todo);
      const int s0 = __builtin_amdgcn_readlane(s, leader);
      const unsigned long long mask = __ballot(live && s == s0);
      const int start = cur[s0];
      if (live && s == s0) {
        const int pos = start + __popcll(mask & ((1ull << lane) - 1ull));
        ord[pos] = e;
        sr[pos] = s0;
      }
      if (lane == leader) cur[s0] = start + __popcll(mask);
      __builtin_amdgcn_wave_barrier();
      todo &= ~mask;
    }
  }
}

// channels 0..2 of QueryAndGroup's output: (xyz[idx] - centre) (/ radius), straight from the
// (B,N,3) point array (the reference transposes it, groups it, subtracts, divides and
// concatenates: group_points.py:100-118)
__global__ __launch_bounds__(GG_BLOCK) void group_xyz_kernel(
    int n, int npoints, int nsample, long long gstride, float radius,
    const float *__restrict__ xyz, const float *__restrict__ centres,
    const int *__restrict__ idx, float *__restrict__ out) {
  const int e = blockIdx.x * GG_BLOCK + threadIdx.x;
  const int bi = blockIdx.y;
  const int e_total = npoints * nsample;
  if (e >= e_total) return;
  int src = idx[(size_t)bi * e_total + e];
  src = src < 0 ? 0 : (src >= n ? n - 1 : src);
  const float *p = xyz + ((size_t)bi * n + src) * 3;
  const float *cc = centres + ((size_t)bi * npoints + e / nsample) * 3;
  float *o = out + (size_t)bi * gstride + e;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    float v = p[d] - cc[d];
    if (radius > 0.f) v = v / radius;
    o[(size_t)d * e_total] = v;
  }
}

// centres[b][m][:] = xyz[b][sample[b][m]][:]: the sampled centres straight from the (B, N, 3) array
// (the reference transposes to channel-major, gathers, transposes back: point_sa_module.py:122-131)
__global__ __launch_bounds__(GG_BLOCK) void gather_rows3_kernel(
    int n, int m, const float *__restrict__ xyz, const int *__restrict__ sample,
    float *__restrict__ centres) {
  const int i = blockIdx.x * GG_BLOCK + threadIdx.x, bi = blockIdx.y;
  if (i >= m) return;
  int s = sample[(size_t)bi * m + i];
  s = s < 0 ? 0 : (s >= n ? n - 1 : s);
  const float *p = xyz + ((size_t)bi * n + s) * 3;
  float *o = centres + ((size_t)bi * m + i) * 3;
  o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
}

// Coordinate gradient of QueryAndGroup when the grouped coordinates are NETWORK OUTPUTS (vote
// aggregation groups the predicted votes around centres sampled from them): channels 0..2 of the
// grouped tensor are (xyz[idx] - centre) / r with centre = xyz[sample], so
//   d_xyz[p] = (1/r) sum_{entries e with idx[e] = p} g[:, e]
//              - (1/r) sum_{m: sample[m] = p} sum_s g[:, m, s]  +  sum_{m: sample[m] = p} d_centre[m]
// One thread per point: its run of the inverted index (found by bisection in the sorted source
// list, walked in ascending column order), then the centres that are this point.  Every point is
// written once, sums in a fixed order: reproducible (the reference reaches the same sums through
// autograd's cat / sub / div / transpose backward and two atomicAdd scatters).
__global__ __launch_bounds__(GG_BLOCK) void qg_xyz_bwd_kernel(
    int n, int m, int ns, long long gstride, float radius, const float *__restrict__ grad_out,
    const int *__restrict__ order, const int *__restrict__ src, const int *__restrict__ sample,
    const float *__restrict__ d_centre, float *__restrict__ d_xyz) {
  const int p = blockIdx.x * GG_BLOCK + threadIdx.x, bi = blockIdx.y;
  if (p >= n) return;
  const int e_total = m * ns;
  const int *sb = src + (size_t)bi * e_total, *ob = order + (size_t)bi * e_total;
  const float *g = grad_out + (size_t)bi * gstride;       // channels 0..2: rows of e_total
  int lo = 0, hi = e_total;                               // first sorted entry with source >= p
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (sb[mid] < p) lo = mid + 1; else hi = mid;
  }
  float a0 = 0.f, a1 = 0.f, a2 = 0.f;
  for (int j = lo; j < e_total && sb[j] == p; ++j) {
    const int e = ob[j];
    a0 += g[e]; a1 += g[(size_t)e_total + e]; a2 += g[2 * (size_t)e_total + e];
  }
  float c0 = 0.f, c1 = 0.f, c2 = 0.f, t0 = 0.f, t1 = 0.f, t2 = 0.f;
  const int *sm = sample + (size_t)bi * m;
  for (int i = 0; i < m; ++i) {
    int s = sm[i];
    s = s < 0 ? 0 : (s >= n ? n - 1 : s);
    if (s != p) continue;
    for (int k = 0; k < ns; ++k) {
      const size_t e = (size_t)i * ns + k;
      c0 += g[e]; c1 += g[(size_t)e_total + e]; c2 += g[2 * (size_t)e_total + e];
    }
    if (d_centre) {
      const float *dc = d_centre + ((size_t)bi * m + i) * 3;
      t0 += dc[0]; t1 += dc[1]; t2 += dc[2];
    }
  }
  a0 -= c0; a1 -= c1; a2 -= c2;
  if (radius > 0.f) { a0 = a0 / radius; a1 = a1 / radius; a2 = a2 / radius; }
  float *o = d_xyz + ((size_t)bi * n + p) * 3;
  o[0] = a0 + t0; o[1] = a1 + t1; o[2] = a2 + t2;
}

}  // namespace nesie

using namespace nesie;

extern "C" int nesie_gather_rows3(int b, int n, int m, const float *xyz, const int *sample,
                                  float *centres, void *stream) {
  const char *W = "gather_rows3";
  NESIE_REQUIRE(b >= 0 && n >= 0 && m >= 0 && b <= 65535, W);
  if (b == 0 || m == 0) return NESIE_OK;
  NESIE_REQUIRE(n >= 1 && xyz && sample && centres, W);
  hipLaunchKernelGGL(gather_rows3_kernel, dim3(cdiv(m, GG_BLOCK), b), dim3(GG_BLOCK), 0,
                     (hipStream_t)stream, n, m, xyz, sample, centres);
  return check_launch(W);
}

extern "C" int nesie_query_and_group_backward_xyz(int b, int c, int n, int npoints, int nsample,
                                                  float radius, const float *grad_out,
                                                  const int *order, const int *sources,
                                                  const int *sample, const float *d_centres,
                                                  float *d_xyz, void *stream) {
  const char *W = "query_and_group_backward_xyz";
  NESIE_REQUIRE(b >= 0 && c >= 0 && n >= 0 && npoints >= 0 && nsample >= 0 && b <= 65535, W);
  if (b == 0 || n == 0) return NESIE_OK;
  NESIE_REQUIRE(grad_out && order && sources && sample && d_xyz, W);
  NESIE_REQUIRE((long long)npoints * nsample < (1ll << 31), W);
  hipLaunchKernelGGL(qg_xyz_bwd_kernel, dim3(cdiv(n, GG_BLOCK), b), dim3(GG_BLOCK), 0,
                     (hipStream_t)stream, n, npoints, nsample,
                     (long long)(3 + c) * npoints * nsample, radius, grad_out, order, sources,
                     sample, d_centres, d_xyz);
  return check_launch(W);
}

extern "C" int nesie_query_and_group_forward(int b, int c, int n, int npoints, int nsample,
                                             const float *xyz, const float *centres,
                                             const float *features, const int *idx,
                                             float radius, float *out, void *stream) {
  const char *W = "query_and_group_forward";
  NESIE_REQUIRE(b >= 0 && c >= 0 && n >= 0 && npoints >= 0 && nsample >= 0, W);
  const long long e_total = (long long)npoints * nsample;
  if (b == 0 || e_total == 0) return NESIE_OK;
  NESIE_REQUIRE(n >= 1 && xyz && centres && idx && out && (c == 0 || features), W);
  NESIE_REQUIRE(e_total < (1ll << 31) && b <= 65535, W);
  const long long gstride = (long long)(3 + c) * e_total;
  hipLaunchKernelGGL(group_xyz_kernel, dim3(cdiv(e_total, GG_BLOCK), b), dim3(GG_BLOCK), 0,
                     (hipStream_t)stream, n, npoints, nsample, gstride, radius, xyz, centres, idx,
                     out);
  int st = check_launch(W);
  if (st || c == 0) return st;
  return launch_group(true, W, b, c, n, e_total, features, idx, out + 3 * e_total, stream,
                      gstride);
}

extern "C" int nesie_query_and_group_backward(int b, int c, int n, int npoints, int nsample,
                                              const float *grad_out, const int *idx,
                                              float *grad_features, void *stream) {
  const char *W = "query_and_group_backward";
  NESIE_REQUIRE(npoints >= 0 && nsample >= 0 && c >= 1, W);
  const long long e_total = (long long)npoints * nsample;
  return launch_group(false, W, b, c, n, e_total, grad_out ? grad_out + 3 * e_total : nullptr,
                      idx, grad_features, stream, (long long)(3 + c) * e_total);
}

extern "C" int nesie_group_points_forward(int b, int c, int n, int npoints, int nsample,
                                          const float *points, const int *idx,
                                          float *out, void *stream) {
  NESIE_REQUIRE(npoints >= 0 && nsample >= 0, "group_points_forward");
  return launch_group(true, "group_points_forward", b, c, n,
                      (long long)npoints * nsample, points, idx, out, stream);
}

extern "C" int nesie_group_points_backward(int b, int c, int n, int npoints, int nsample,
                                           const float *grad_out, const int *idx,
                                           float *grad_points, void *stream) {
  NESIE_REQUIRE(npoints >= 0 && nsample >= 0, "group_points_backward");
  return launch_group(false, "group_points_backward", b, c, n,
                      (long long)npoints * nsample, grad_out, idx, grad_points, stream);
}

extern "C" int nesie_gather_points_wrapper(int b, int c, int n, int npoints,
                                           const float *points, const int *idx,
                                           float *out, void *stream) {
  NESIE_REQUIRE(npoints >= 0, "gather_points_wrapper");
  return launch_group(true, "gather_points_wrapper", b, c, n, npoints, points, idx, out,
                      stream);
}

extern "C" int nesie_gather_points_grad_wrapper(int b, int c, int n, int npoints,
                                                const float *grad_out, const int *idx,
                                                float *grad_points, void *stream) {
  NESIE_REQUIRE(npoints >= 0, "gather_points_grad_wrapper");
  return launch_group(false, "gather_points_grad_wrapper", b, c, n, npoints, grad_out,
                      idx, grad_points, stream);
}

extern "C" int nesie_query_and_group_backward_csr(int b, int c, int n, int npoints, int nsample,
                                                  const float *grad_out, const int *order,
                                                  const int *sources, float *grad_features,
                                                  void *stream) {
  const char *W = "query_and_group_backward_csr";
  NESIE_REQUIRE(b >= 0 && c >= 1 && n >= 0 && npoints >= 0 && nsample >= 0, W);
  const long long e_total = (long long)npoints * nsample;
  if (b == 0 || n == 0) return NESIE_OK;
  NESIE_REQUIRE(grad_out && order && sources && grad_features, W);
  if (e_total == 0) return NESIE_OK;
  NESIE_REQUIRE(e_total < (1ll << 31) && b <= 65535 && cdiv(c, GG_CH) <= 65535, W);
  if (launch_group_bwd_csr_rows(b, c, n, e_total, (long long)(3 + c) * e_total, 1, grad_out + 3 * e_total, nullptr,
                                order, sources, grad_features, (hipStream_t)stream))
    return check_launch(W);
  zero_fill(grad_features, (long long)b * c * n, (hipStream_t)stream);      // (the HBM-gather form ADDS)
  hipLaunchKernelGGL(group_bwd_csr_kernel, dim3(cdiv(e_total, GG_BLOCK), cdiv(c, GG_CH), b),
                     dim3(GG_BLOCK), 0, (hipStream_t)stream, c, n, (int)e_total,
                     (long long)(3 + c) * e_total, 1, grad_out + 3 * e_total,
                     (const float *)nullptr, order, sources, grad_features);
  return check_launch(W);
}

// group_points / gather_points backward through an inverted index of idx (nesie_inverted_index):
// what nesie_group_points_backward adds with float atomics (LDS or HBM), in an order fixed by the
// index alone.  grad_points must be zero on entry.
extern "C" int nesie_group_points_backward_csr(int b, int c, int n, int npoints, int nsample,
                                               const float *grad_out, const int *order,
                                               const int *sources, float *grad_points, void *stream) {
  const char *W = "group_points_backward_csr";
  NESIE_REQUIRE(b >= 0 && c >= 0 && n >= 0 && npoints >= 0 && nsample >= 0, W);
  const long long e_total = (long long)npoints * nsample;
  if (b == 0 || n == 0 || c == 0 || e_total == 0) return NESIE_OK;
  NESIE_REQUIRE(grad_out && order && sources && grad_points, W);
  NESIE_REQUIRE(e_total < (1ll << 31) && b <= 65535 && cdiv(c, GG_CH) <= 65535, W);
  if (launch_group_bwd_csr_rows(b, c, n, e_total, (long long)c * e_total, 1, grad_out, nullptr, order, sources,
                                grad_points, (hipStream_t)stream))
    return check_launch(W);
  zero_fill(grad_points, (long long)b * c * n, (hipStream_t)stream);
  hipLaunchKernelGGL(group_bwd_csr_kernel, dim3(cdiv(e_total, GG_BLOCK), cdiv(c, GG_CH), b),
                     dim3(GG_BLOCK), 0, (hipStream_t)stream, c, n, (int)e_total, (long long)c * e_total, 1,
                     grad_out, (const float *)nullptr, order, sources, grad_points);
  return check_launch(W);
}

namespace nesie {
int launch_inverted_index(int b, int n, long long e_total, const int *idx, int *order, int *sources,
                          int *scratch, hipStream_t s) {
  const char *W = "inverted_index";
  NESIE_REQUIRE(b >= 0 && n >= 1 && e_total >= 0 && e_total < (1ll << 31), W);
  if (b == 0) return NESIE_OK;
  NESIE_REQUIRE(idx && order && sources && scratch, W);
  static const int stable_on = getenv("NESIE_INDEX_STABLE") ? atoi(getenv("NESIE_INDEX_STABLE")) : 1;   // A/B switch
  if (stable_on && n <= II_STABLE_N) {               // stable counting sort: ascending runs without a ranking pass
    const size_t lds = ((size_t)II_WAVES * n + II_BLOCK) * sizeof(int);
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute((const void *)inverted_index_stable_kernel,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)(((size_t)II_WAVES * II_STABLE_N + II_BLOCK) * sizeof(int)));
      attr = true;
    }
    hipLaunchKernelGGL(inverted_index_stable_kernel, dim3(b), dim3(II_BLOCK), lds, s, n, (int)e_total, idx,
                       order, sources);
    return check_launch(W);
  }
  const int windows = cdiv(n, II_WINDOW);
  NESIE_REQUIRE(windows <= 65535, W);
  const size_t lds = ((size_t)(n < II_WINDOW ? n : II_WINDOW) + II_BLOCK + 1) * sizeof(int);
  hipLaunchKernelGGL(inverted_index_kernel, dim3(b, windows), dim3(II_BLOCK), lds, s, n, (int)e_total, idx, order,
                     sources, scratch);
  return check_launch(W);
}
}  // namespace nesie

extern "C" int nesie_inverted_index(int b, int n, long long e_total, const int *idx, int *order,
                                    int *sources, int *scratch, void *stream) {
  return launch_inverted_index(b, n, e_total, idx, order, sources, scratch, (hipStream_t)stream);
}

extern "C" int nesie_three_interpolate_grad_csr(int b, int c, int n, int m, const float *grad_out,
                                                long long grad_out_bstride, const float *weight,
                                                const int *order, const int *sources, float *grad_points,
                                                void *stream) {
  const char *W = "three_interpolate_grad_csr";
  NESIE_REQUIRE(b >= 0 && c >= 1 && n >= 0 && m >= 1, W);
  if (b == 0) return NESIE_OK;
  NESIE_REQUIRE(grad_points, W);
  if (n == 0) {
    zero_fill(grad_points, (long long)b * c * m, (hipStream_t)stream);
    return check_launch(W);
  }
  NESIE_REQUIRE(grad_out && weight && order && sources && grad_out_bstride >= (long long)c * n, W);
  NESIE_REQUIRE((long long)n * 3 < (1ll << 31) && b <= 65535 && cdiv(c, GG_CH) <= 65535, W);
  if (launch_group_bwd_csr_rows(b, c, m, (long long)n * 3, grad_out_bstride, 3, grad_out, weight, order, sources,
                                grad_points, (hipStream_t)stream))
    return check_launch(W);
  zero_fill(grad_points, (long long)b * c * m, (hipStream_t)stream);
  hipLaunchKernelGGL(group_bwd_csr_kernel, dim3(cdiv((long long)n * 3, GG_BLOCK), cdiv(c, GG_CH), b),
                     dim3(GG_BLOCK), 0, (hipStream_t)stream, c, m, n * 3, grad_out_bstride, 3,
                     grad_out, weight, order, sources, grad_points);
  return check_launch(W);
}
