/*
 * nesie_oracle.c -- CPU restatement of the reference's point-cloud operators.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under nesie_amd/ may import, link or call
 * this file.  It is used by tests/, by __graft_entry__.smoke() and by the
 * cpu_baseline leg of bench.py, always as the checker / the timed CPU
 * baseline, never as the product path.
 *
 * What it restates (reference = /root/reference, mmdetection3d-0.x fork):
 *   mmdet3d/ops/furthest_point_sample/src/furthest_point_sample_cuda.cu
 *   mmdet3d/ops/ball_query/src/ball_query_cuda.cu
 *   mmdet3d/ops/group_points/src/group_points_cuda.cu
 *   mmdet3d/ops/gather_points/src/gather_points_cuda.cu
 *   mmdet3d/ops/interpolate/src/three_nn_cuda.cu, three_interpolate_cuda.cu
 *   mmdet3d/ops/rotated_iou/cuda_op/sort_vert_kernel.cu
 *   mmdet3d/ops/roiaware_pool3d/src/points_in_boxes_cuda.cu
 *
 * PIN STATUS.  The reference ships these operators as CUDA only, with no
 * tests, fixtures or golden vectors, and it
 * cannot run in this image.  So for fps / ball_query / group / gather /
 * three_nn / three_interpolate / sort_vertices the parity is UNPINNED against
 * reference outputs; the restatement is instead cross-checked in
 * tests/test_oracle.py against (i) an independent numpy restatement and
 * (ii) for FPS a literal thread-by-thread simulation of the reference's
 * LDS tree reduction.  points_in_boxes is PINNED: it is checked against
 * oracle/_ref/points_in_boxes_ref.so, compiled from the reference's own
 * roiaware_pool3d/src/points_in_boxes_cpu.cpp by oracle/Makefile.
 *
 * Floating point discipline: every squared distance
 * is ((dx*dx) + (dy*dy)) + (dz*dz) in fp32 with NO fma contraction; build
 * with -ffp-contract=off (oracle/Makefile does).
 *
 * Distance-form switch (default 0; also settable at build time with
 * -DORACLE_FMA_DIST=<form>).  nvcc's default -fmad=true MAY contract the
 * reference's `(x2-x1)*(x2-x1) + (y2-y1)*(y2-y1) + (z2-z1)*(z2-z1)`
 * (furthest_point_sample_cuda.cu:65-66, ball_query_cuda.cu:41-42,
 * three_nn_cuda.cu:41) into a fused chain; which one cannot be known without
 * running the reference's own CUDA build.  oracle_set_distance_form() selects
 *   0  ((dx*dx) + (dy*dy)) + (dz*dz)              no contraction (the product's form)
 *   1  fma(dz, dz, fma(dx, dx, dy*dy))             LLVM's contraction of the expression as
 *                                                  written: the LEFT product of a sum of two
 *                                                  products is fused, then the outer add
 *   2  fma(dz, dz, fma(dy, dy, dx*dx))             the other choice for the inner sum
 * with correctly rounded fmaf(), so that the day a CUDA-produced fixture exists
 * the contraction question is a one-flag check of oracle AND HIP kernels
 * (nesie_set_distance_form, same numbering) instead of a rewrite.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef ORACLE_FMA_DIST
#define ORACLE_FMA_DIST 0
#endif
static int g_dist_form = ORACLE_FMA_DIST;

int oracle_set_distance_form(int form) {
  if (form < 0 || form > 2) return -1;
  g_dist_form = form;
  return 0;
}
int oracle_get_distance_form(void) { return g_dist_form; }

/* the squared distance of SURVEY.md appendix A.0 in the selected form */
static inline float sqdist(float dx, float dy, float dz, int form) {
  if (form == 1) return fmaf(dz, dz, fmaf(dx, dx, dy * dy));
  if (form == 2) return fmaf(dz, dz, fmaf(dy, dy, dx * dx));
  return ((dx * dx) + (dy * dy)) + (dz * dz);
}

/* ------------------------------------------------------------------ */
/* FPS                                                                 */
/* ------------------------------------------------------------------ */

/* furthest_point_sample_cuda.cu:11-15  opt_n_threads() */
int oracle_fps_block_size(int n) {
  int pow_2 = (int)(log((double)n) / log(2.0));
  int bs = 1 << pow_2;
  if (bs > 1024) bs = 1024;
  if (bs < 1) bs = 1;
  return bs;
}

/* furthest_point_sample_cuda.cu:17-23 (__update) and :76-136 (tree) */
static int fps_tree_reduce(float *dists, int *dists_i, int bs) {
  for (int stride = bs / 2; stride >= 1; stride >>= 1) {
    for (int t = 0; t < stride; ++t) {
      float v1 = dists[t], v2 = dists[t + stride];
      int i1 = dists_i[t], i2 = dists_i[t + stride];
      dists[t] = v1 > v2 ? v1 : v2; /* max(v1, v2) */
      dists_i[t] = v2 > v1 ? i2 : i1;
    }
  }
  return dists_i[0];
}

/* furthest_point_sample_cuda.cu:26-141  furthest_point_sampling_kernel.
 * xyz (B,N,3) f32, temp (B,N) f32 (caller fills 1e10; updated in place),
 * idx (B,M) i32.  Thread t of the reference owns k == t (mod bs) and keeps
 * the first strict maximum; the tree then resolves ties between threads. */
int oracle_furthest_point_sampling(int b, int n, int m, const float *xyz,
                                   float *temp, int *idx) {
  if (m <= 0) return 1; /* :34 */
  const int form = g_dist_form;
  const int bs = oracle_fps_block_size(n);
#pragma omp parallel for schedule(static)
  for (int bi = 0; bi < b; ++bi) {
    const float *p = xyz + (size_t)bi * n * 3;
    float *tp = temp + (size_t)bi * n;
    int *out = idx + (size_t)bi * m;
    float *d2buf = (float *)malloc(sizeof(float) * (size_t)n);
    float *dists = (float *)malloc(sizeof(float) * (size_t)bs);
    int *dists_i = (int *)malloc(sizeof(int) * (size_t)bs);
    int old = 0;
    out[0] = 0; /* :47 */
    for (int j = 1; j < m; ++j) {
      const float x1 = p[old * 3 + 0], y1 = p[old * 3 + 1], z1 = p[old * 3 + 2];
      for (int k = 0; k < n; ++k) { /* :58-71, vectorisable part */
        float dx = p[k * 3 + 0] - x1;
        float dy = p[k * 3 + 1] - y1;
        float dz = p[k * 3 + 2] - z1;
        float d = sqdist(dx, dy, dz, form);
        float t = tp[k];
        float d2 = d < t ? d : t; /* min(d, temp[k]) */
        tp[k] = d2;
        d2buf[k] = d2;
      }
      for (int t = 0; t < bs; ++t) { /* :52-53 */
        dists[t] = -1.0f;
        dists_i[t] = 0;
      }
      for (int k = 0; k < n; ++k) { /* :69-70 per-thread first strict max */
        int t = k & (bs - 1);
        if (d2buf[k] > dists[t]) {
          dists[t] = d2buf[k];
          dists_i[t] = k;
        }
      }
      old = fps_tree_reduce(dists, dists_i, bs);
      out[j] = old; /* :138-139 */
    }
    free(d2buf);
    free(dists);
    free(dists_i);
  }
  return 1;
}

/* furthest_point_sample_cuda.cu:214-331  ..._with_dist_kernel (F-FPS).
 * dist (B,N,N); same reduction, d = dist[old*n + k]. */
int oracle_furthest_point_sampling_with_dist(int b, int n, int m,
                                             const float *dist, float *temp,
                                             int *idx) {
  if (m <= 0) return 1;
  const int bs = oracle_fps_block_size(n);
#pragma omp parallel for schedule(static)
  for (int bi = 0; bi < b; ++bi) {
    const float *dm = dist + (size_t)bi * n * n;
    float *tp = temp + (size_t)bi * n;
    int *out = idx + (size_t)bi * m;
    float *dists = (float *)malloc(sizeof(float) * (size_t)bs);
    int *dists_i = (int *)malloc(sizeof(int) * (size_t)bs);
    int old = 0;
    out[0] = 0;
    for (int j = 1; j < m; ++j) {
      for (int t = 0; t < bs; ++t) {
        dists[t] = -1.0f;
        dists_i[t] = 0;
      }
      for (int k = 0; k < n; ++k) {
        float d = dm[(size_t)old * n + k];
        float t0 = tp[k];
        float d2 = d < t0 ? d : t0;
        tp[k] = d2;
        int t = k & (bs - 1);
        if (d2 > dists[t]) {
          dists[t] = d2;
          dists_i[t] = k;
        }
      }
      old = fps_tree_reduce(dists, dists_i, bs);
      out[j] = old;
    }
    free(dists);
    free(dists_i);
  }
  return 1;
}

/* ------------------------------------------------------------------ */
/* ball query                                                          */
/* ------------------------------------------------------------------ */

/* ball_query_cuda.cu:11-54.  new_xyz (B,M,3), xyz (B,N,3), idx (B,M,ns) i32
 * zero-initialised by the caller (ball_query.py:35). */
int oracle_ball_query(int b, int n, int m, float min_radius, float max_radius,
                      int nsample, const float *new_xyz, const float *xyz,
                      int *idx) {
  const int form = g_dist_form;
  const float max_radius2 = max_radius * max_radius; /* :27 */
  const float min_radius2 = min_radius * min_radius; /* :28 */
#pragma omp parallel for collapse(2) schedule(static)
  for (int bi = 0; bi < b; ++bi) {
    for (int pi = 0; pi < m; ++pi) {
      const float *c = new_xyz + ((size_t)bi * m + pi) * 3;
      const float *p = xyz + (size_t)bi * n * 3;
      int *o = idx + ((size_t)bi * m + pi) * nsample;
      const float new_x = c[0], new_y = c[1], new_z = c[2];
      int cnt = 0;
      for (int k = 0; k < n; ++k) {
        float dx = new_x - p[k * 3 + 0];
        float dy = new_y - p[k * 3 + 1];
        float dz = new_z - p[k * 3 + 2];
        float d2 = sqdist(dx, dy, dz, form); /* :41-42 */
        if (d2 == 0 || (d2 >= min_radius2 && d2 < max_radius2)) { /* :43 */
          if (cnt == 0) {
            for (int l = 0; l < nsample; ++l) o[l] = k; /* :44-48 */
          }
          o[cnt] = k;
          ++cnt;
          if (cnt >= nsample) break;
        }
      }
    }
  }
  return 1;
}

/* ------------------------------------------------------------------ */
/* group / gather                                                      */
/* ------------------------------------------------------------------ */

/* group_points_cuda.cu:56-80.  points (B,C,N), idx (B,M,ns) -> out (B,C,M,ns) */
int oracle_group_points(int b, int c, int n, int npoints, int nsample,
                        const float *points, const int *idx, float *out) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int bi = 0; bi < b; ++bi)
    for (int ci = 0; ci < c; ++ci) {
      const float *src = points + ((size_t)bi * c + ci) * n;
      const int *ix = idx + (size_t)bi * npoints * nsample;
      float *dst = out + ((size_t)bi * c + ci) * npoints * nsample;
      for (int e = 0; e < npoints * nsample; ++e) dst[e] = src[ix[e]];
    }
  return 1;
}

/* group_points_cuda.cu:10-31 (atomicAdd scatter).  grad_points must be
 * zeroed by the caller.  Summation order here: ascending (p, s). */
int oracle_group_points_grad(int b, int c, int n, int npoints, int nsample,
                             const float *grad_out, const int *idx,
                             float *grad_points) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int bi = 0; bi < b; ++bi)
    for (int ci = 0; ci < c; ++ci) {
      float *dst = grad_points + ((size_t)bi * c + ci) * n;
      const int *ix = idx + (size_t)bi * npoints * nsample;
      const float *g = grad_out + ((size_t)bi * c + ci) * npoints * nsample;
      for (int e = 0; e < npoints * nsample; ++e) dst[ix[e]] += g[e];
    }
  return 1;
}

/* gather_points_cuda.cu:8-26.  points (B,C,N), idx (B,M) -> out (B,C,M) */
int oracle_gather_points(int b, int c, int n, int npoints, const float *points,
                         const int *idx, float *out) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int bi = 0; bi < b; ++bi)
    for (int ci = 0; ci < c; ++ci) {
      const float *src = points + ((size_t)bi * c + ci) * n;
      const int *ix = idx + (size_t)bi * npoints;
      float *dst = out + ((size_t)bi * c + ci) * npoints;
      for (int e = 0; e < npoints; ++e) dst[e] = src[ix[e]];
    }
  return 1;
}

/* gather_points_cuda.cu:51-70 (atomicAdd scatter), ascending m order. */
int oracle_gather_points_grad(int b, int c, int n, int npoints,
                              const float *grad_out, const int *idx,
                              float *grad_points) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int bi = 0; bi < b; ++bi)
    for (int ci = 0; ci < c; ++ci) {
      float *dst = grad_points + ((size_t)bi * c + ci) * n;
      const int *ix = idx + (size_t)bi * npoints;
      const float *g = grad_out + ((size_t)bi * c + ci) * npoints;
      for (int e = 0; e < npoints; ++e) dst[ix[e]] += g[e];
    }
  return 1;
}

/* ------------------------------------------------------------------ */
/* three_nn / three_interpolate                                        */
/* ------------------------------------------------------------------ */

/* three_nn_cuda.cu:11-65.  unknown (B,n,3), known (B,m,3) ->
 * dist2 (B,n,3) f32, idx (B,n,3) i32.  bests are double, d is float. */
int oracle_three_nn(int b, int n, int m, const float *unknown,
                    const float *known, float *dist2, int *idx) {
  const int form = g_dist_form;
#pragma omp parallel for collapse(2) schedule(static)
  for (int bi = 0; bi < b; ++bi)
    for (int pi = 0; pi < n; ++pi) {
      const float *u = unknown + ((size_t)bi * n + pi) * 3;
      const float *kn = known + (size_t)bi * m * 3;
      const float ux = u[0], uy = u[1], uz = u[2];
      double best1 = 1e40, best2 = 1e40, best3 = 1e40; /* :35 */
      int besti1 = 0, besti2 = 0, besti3 = 0;
      for (int k = 0; k < m; ++k) {
        float dx = ux - kn[k * 3 + 0];
        float dy = uy - kn[k * 3 + 1];
        float dz = uz - kn[k * 3 + 2];
        float d = sqdist(dx, dy, dz, form); /* :41 */
        if (d < best1) {
          best3 = best2; besti3 = besti2;
          best2 = best1; besti2 = besti1;
          best1 = d; besti1 = k;
        } else if (d < best2) {
          best3 = best2; besti3 = besti2;
          best2 = d; besti2 = k;
        } else if (d < best3) {
          best3 = d; besti3 = k;
        }
      }
      float *od = dist2 + ((size_t)bi * n + pi) * 3;
      int *oi = idx + ((size_t)bi * n + pi) * 3;
      od[0] = (float)best1; od[1] = (float)best2; od[2] = (float)best3;
      oi[0] = besti1; oi[1] = besti2; oi[2] = besti3;
    }
  return 1;
}

/* three_interpolate_cuda.cu:11-35.  points (B,C,M), idx/weight (B,N,3) ->
 * out (B,C,N); products rounded individually, summed left to right. */
int oracle_three_interpolate(int b, int c, int m, int n, const float *points,
                             const int *idx, const float *weight, float *out) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int bi = 0; bi < b; ++bi)
    for (int ci = 0; ci < c; ++ci) {
      const float *src = points + ((size_t)bi * c + ci) * m;
      const int *ix = idx + (size_t)bi * n * 3;
      const float *w = weight + (size_t)bi * n * 3;
      float *dst = out + ((size_t)bi * c + ci) * n;
      for (int p = 0; p < n; ++p) {
        float a0 = w[p * 3 + 0] * src[ix[p * 3 + 0]];
        float a1 = w[p * 3 + 1] * src[ix[p * 3 + 1]];
        float a2 = w[p * 3 + 2] * src[ix[p * 3 + 2]];
        dst[p] = (a0 + a1) + a2;
      }
    }
  return 1;
}

/* three_interpolate_cuda.cu:61-84 (3 atomicAdds per element).
 * grad_out (B,C,N) -> grad_points (B,C,M) zeroed by the caller. */
int oracle_three_interpolate_grad(int b, int c, int n, int m,
                                  const float *grad_out, const int *idx,
                                  const float *weight, float *grad_points) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int bi = 0; bi < b; ++bi)
    for (int ci = 0; ci < c; ++ci) {
      float *dst = grad_points + ((size_t)bi * c + ci) * m;
      const int *ix = idx + (size_t)bi * n * 3;
      const float *w = weight + (size_t)bi * n * 3;
      const float *g = grad_out + ((size_t)bi * c + ci) * n;
      for (int p = 0; p < n; ++p) {
        dst[ix[p * 3 + 0]] += g[p] * w[p * 3 + 0];
        dst[ix[p * 3 + 1]] += g[p] * w[p * 3 + 1];
        dst[ix[p * 3 + 2]] += g[p] * w[p * 3 + 2];
      }
    }
  return 1;
}

/* ------------------------------------------------------------------ */
/* sort_vertices                                                       */
/* ------------------------------------------------------------------ */

#define SV_MAX_IDX 9
#define SV_INTER_OFFSET 8
#define SV_EPS 1e-8 /* a double constant, as in the reference */

/* sort_vert_kernel.cu:15-40.  The reference falls off the end of this
 * function (undefined behaviour) when exactly one of y1, y2 is 0 or both are;
 * this restatement returns false there, and tests avoid such inputs. */
static int sv_compare(float x1, float y1, float x2, float y2) {
  /* fabs on a float is the float overload in the reference's C++ */
  if (fabsf(x1 - x2) < SV_EPS && fabsf(y2 - y1) < SV_EPS) return 0;
  if (y1 > 0 && y2 < 0) return 1;
  if (y1 < 0 && y2 > 0) return 0;
  float n1 = x1 * x1 + y1 * y1 + SV_EPS;
  float n2 = x2 * x2 + y2 * y2 + SV_EPS;
  if (y1 > 0 && y2 > 0) {
    return (fabsf(x1) * x1 / n1 - fabsf(x2) * x2 / n2 > SV_EPS) ? 1 : 0;
  }
  if (y1 < 0 && y2 < 0) {
    return (fabsf(x1) * x1 / n1 - fabsf(x2) * x2 / n2 < SV_EPS) ? 1 : 0;
  }
  return 0;
}

/* sort_vert_kernel.cu:42-134.  vertices (B,N,M,2) f32, mask (B,N,M) u8,
 * num_valid (B,N) i32 -> idx (B,N,9) i32.  `pad` is uninitialised in the
 * reference when no invalid intersection exists; here it starts at M-1. */
int oracle_sort_vertices(int b, int n, int m, const float *vertices,
                         const uint8_t *mask, const int *num_valid, int *idx) {
#pragma omp parallel for schedule(static)
  for (long long bn = 0; bn < (long long)b * n; ++bn) {
    const float *v = vertices + (size_t)bn * m * 2;
    const uint8_t *mk = mask + (size_t)bn * m;
    const int nv = num_valid[bn];
    int *o = idx + (size_t)bn * SV_MAX_IDX;
    int pad = m - 1;
    for (int j = SV_INTER_OFFSET; j < m; ++j) {
      if (!mk[j]) { pad = j; break; }
    }
    if (nv < 3) {
      for (int j = 0; j < SV_MAX_IDX; ++j) o[j] = pad;
      continue;
    }
    for (int j = 0; j < nv; ++j) {
      float x_min = 1;
      float y_min = -SV_EPS;
      int i_take = 0;
      for (int k = 0; k < m; ++k) {
        float x = v[k * 2 + 0], y = v[k * 2 + 1];
        if (j == 0) {
          if (mk[k] && sv_compare(x, y, x_min, y_min)) {
            x_min = x; y_min = y; i_take = k;
          }
        } else {
          int i2 = o[j - 1];
          float x2 = v[i2 * 2 + 0], y2 = v[i2 * 2 + 1];
          if (mk[k] && sv_compare(x, y, x_min, y_min) &&
              sv_compare(x2, y2, x, y)) {
            x_min = x; y_min = y; i_take = k;
          }
        }
        o[j] = i_take;
      }
    }
    o[nv] = o[0];
    for (int j = nv + 1; j < SV_MAX_IDX; ++j) o[j] = pad;
    if (nv == 8) { /* :114-129 identical boxes */
      int counter = 0;
      for (int j = 0; j < 4; ++j) {
        int check = o[j];
        for (int k = 4; k < SV_INTER_OFFSET; ++k)
          if (o[k] == check) counter++;
      }
      if (counter == 4) {
        o[4] = o[0];
        for (int j = 5; j < SV_MAX_IDX; ++j) o[j] = pad;
      }
    }
  }
  return 1;
}

/* ------------------------------------------------------------------ */
/* points_in_boxes_batch                                               */
/* ------------------------------------------------------------------ */

/* points_in_boxes_cuda.cu:24-49.  cos/sin are taken in double and rounded to
 * float (the canonical form shared with the HIP kernel; the reference's
 * device cosf/sinf differ from it by at most their own 2-ulp error). */
static int pib_check(const float *pt, const float *box3d) {
  float x = pt[0], y = pt[1], z = pt[2];
  float cx = box3d[0], cy = box3d[1], cz = box3d[2];
  float w = box3d[3], l = box3d[4], h = box3d[5], rz = box3d[6];
  cz += h / 2.0; /* double division, rounded to float on the += */
  if (fabsf(z - cz) > h / 2.0) return 0;
  float rot_angle = rz + M_PI / 2;
  float cosa = (float)cos((double)rot_angle), sina = (float)sin((double)rot_angle);
  float shift_x = x - cx, shift_y = y - cy;
  float local_x = shift_x * cosa + shift_y * (-sina);
  float local_y = shift_x * sina + shift_y * cosa;
  return (local_x > -l / 2.0) & (local_x < l / 2.0) & (local_y > -w / 2.0) &
         (local_y < w / 2.0);
}

/* points_in_boxes_cuda.cu:79-105.  boxes (B,T,7) LiDAR frame bottom-centre,
 * pts (B,M,3) -> out (B,M,T) i32 zeroed by the caller; writes 1 where inside. */
int oracle_points_in_boxes_batch(int b, int boxes_num, int pts_num,
                                 const float *boxes, const float *pts, int *out) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int bi = 0; bi < b; ++bi)
    for (int pi = 0; pi < pts_num; ++pi) {
      const float *pt = pts + ((size_t)bi * pts_num + pi) * 3;
      const float *bx = boxes + (size_t)bi * boxes_num * 7;
      int *o = out + ((size_t)bi * pts_num + pi) * boxes_num;
      for (int k = 0; k < boxes_num; ++k)
        if (pib_check(pt, bx + k * 7)) o[k] = 1;
    }
  return 1;
}

int oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ------------------------------------------------------------------ */
/* LHS-NMS of pseudo labels                                            */
/* ------------------------------------------------------------------ */

/* lhs_3d_faster_samecls (mmdet3d/models/detectors/votenet_nesie.py:733-779), the
 * numpy routine the teacher's pseudo boxes go through: greedy over descending score;
 * the boxes that overlap the current pick (IoU > thr, same class) are removed, but the
 * better-scored HALF of them is kept as well.  boxes (B,K,8) f32 = axis-aligned
 * (x1,y1,z1,x2,y2,z2,score,class); keep (B,K) u8 = 1 for the reference's `pick`.
 * Arithmetic in double like numpy's float64 arrays.  np.argsort leaves the order of
 * equal scores unspecified; here (and in the HIP kernel) ties order by index. */
int oracle_lhs_nms_samecls(int b, int k, const float *boxes, float thr, uint8_t *keep) {
  for (int bi = 0; bi < b; ++bi) {
    const float *bx = boxes + (size_t)bi * k * 8;
    uint8_t *kp = keep + (size_t)bi * k;
    int *order = (int *)malloc(sizeof(int) * (size_t)k); /* ascending (score, index) */
    int n = k;
    for (int i = 0; i < k; ++i) { order[i] = i; kp[i] = 0; }
    for (int i = 1; i < k; ++i) { /* stable insertion sort */
      int v = order[i], j = i - 1;
      while (j >= 0 && bx[order[j] * 8 + 6] > bx[v * 8 + 6]) { order[j + 1] = order[j]; --j; }
      order[j + 1] = v;
    }
    int *ov = (int *)malloc(sizeof(int) * (size_t)k);
    while (n > 0) {
      const int i = order[n - 1];
      kp[i] = 1;
      const double ai = ((double)bx[i*8+3] - bx[i*8+0]) * ((double)bx[i*8+4] - bx[i*8+1]) *
                        ((double)bx[i*8+5] - bx[i*8+2]) + 1e-8;
      int nov = 0;
      for (int p = 0; p < n - 1; ++p) {
        const int j = order[p];
        double xx1 = fmax(bx[i*8+0], bx[j*8+0]), yy1 = fmax(bx[i*8+1], bx[j*8+1]);
        double zz1 = fmax(bx[i*8+2], bx[j*8+2]), xx2 = fmin(bx[i*8+3], bx[j*8+3]);
        double yy2 = fmin(bx[i*8+4], bx[j*8+4]), zz2 = fmin(bx[i*8+5], bx[j*8+5]);
        double l = fmax(0.0, xx2 - xx1), w = fmax(0.0, yy2 - yy1), h = fmax(0.0, zz2 - zz1);
        double aj = ((double)bx[j*8+3] - bx[j*8+0]) * ((double)bx[j*8+4] - bx[j*8+1]) *
                    ((double)bx[j*8+5] - bx[j*8+2]) + 1e-8;
        double inter = l * w * h;
        double o = inter / (ai + aj - inter);
        o = o * (bx[i*8+7] == bx[j*8+7] ? 1.0 : 0.0);
        if (o > (double)thr) ov[nov++] = p; /* positions in `order`, ascending */
      }
      for (int c = 0; c < nov / 2; ++c) kp[order[ov[nov - c - 1]]] = 1;
      /* delete position n-1 and the overlapped positions */
      int w_ = 0, q = 0;
      for (int p = 0; p < n - 1; ++p) {
        if (q < nov && ov[q] == p) { ++q; continue; }
        order[w_++] = order[p];
      }
      n = w_;
    }
    free(order);
    free(ov);
  }
  return 1;
}

/* ------------------------------------------------------------------ */
/* inference post-processing / evaluation geometry    */
/* ------------------------------------------------------------------ */

/* aligned_3d_nms (mmdet3d/core/post_processing/box3d_nms.py:129-176) as the python reads:
 * a list of candidates in ascending score order; take the last, compute fp32 IoU with all
 * the others (times the same-class flag), keep those with iou <= thr.  boxes (B,K,6),
 * scores/classes (B,K), valid (B,K) u8 or NULL (the boolean-mask indexing of the caller,
 * nesie_head.py:752-755); picks (B,K) = positions in pick order, -1 padded; count (B).
 * PIN: tests/golden/inference_golden.pt holds the outputs of the reference function itself
 * (imported by path) on seeded inputs without score ties.  torch.argsort leaves equal scores
 * unordered; here they order by index (stable ascending sort; NaN scores sort last). */
int oracle_aligned_3d_nms(int b, int k, const float *boxes, const float *scores,
                          const int *classes, const uint8_t *valid, float thr, int *picks,
                          int *count) {
  for (int bi = 0; bi < b; ++bi) {
    const float *bx = boxes + (size_t)bi * k * 6;
    const float *sc = scores + (size_t)bi * k;
    const int *cl = classes + (size_t)bi * k;
    int *pk = picks + (size_t)bi * k;
    int *list = (int *)malloc(sizeof(int) * (size_t)(k ? k : 1));
    int n = 0, np = 0;
    for (int i = 0; i < k; ++i) {
      pk[i] = -1;
      if (!valid || valid[(size_t)bi * k + i]) list[n++] = i;
    }
    for (int i = 1; i < n; ++i) { /* stable insertion sort, ascending; NaN = largest */
      int v = list[i], j = i - 1;
      float sv = sc[v];
      while (j >= 0) {
        float sj = sc[list[j]];
        int greater = (sj != sj) ? (sv == sv) : (sv == sv && sj > sv);
        if (!greater) break;
        list[j + 1] = list[j];
        --j;
      }
      list[j + 1] = v;
    }
    while (n > 0) {
      const int i = list[n - 1];
      pk[np++] = i;
      const float ai = (bx[i*6+3] - bx[i*6+0]) * (bx[i*6+4] - bx[i*6+1]) * (bx[i*6+5] - bx[i*6+2]);
      int w = 0;
      for (int p = 0; p < n - 1; ++p) {
        const int j = list[p];
        float xx1 = fmaxf(bx[i*6+0], bx[j*6+0]), yy1 = fmaxf(bx[i*6+1], bx[j*6+1]);
        float zz1 = fmaxf(bx[i*6+2], bx[j*6+2]), xx2 = fminf(bx[i*6+3], bx[j*6+3]);
        float yy2 = fminf(bx[i*6+4], bx[j*6+4]), zz2 = fminf(bx[i*6+5], bx[j*6+5]);
        float l = fmaxf(0.f, xx2 - xx1), wd = fmaxf(0.f, yy2 - yy1), h = fmaxf(0.f, zz2 - zz1);
        float aj = (bx[j*6+3] - bx[j*6+0]) * (bx[j*6+4] - bx[j*6+1]) * (bx[j*6+5] - bx[j*6+2]);
        float inter = l * wd * h;
        float iou = inter / (ai + aj - inter);
        iou = iou * (cl[i] == cl[j] ? 1.f : 0.f);
        if (iou <= thr) list[w++] = j;
      }
      n = w;
    }
    count[bi] = np;
    free(list);
  }
  return 1;
}

/* Column sums of points_in_boxes_batch: counts (B,T) (nesie_head.py:744-750). */
int oracle_points_in_boxes_count(int b, int boxes_num, int pts_num, const float *boxes,
                                 const float *pts, int *counts) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int bi = 0; bi < b; ++bi)
    for (int k = 0; k < boxes_num; ++k) {
      const float *bx = boxes + ((size_t)bi * boxes_num + k) * 7;
      int c = 0;
      for (int pi = 0; pi < pts_num; ++pi) c += pib_check(pts + ((size_t)bi * pts_num + pi) * 3, bx);
      counts[(size_t)bi * boxes_num + k] = c;
    }
  return 1;
}

/* Rotated BEV overlap area, iou3d_kernel.cu:127-238 (box_overlap) with its helpers
 * (:36-125).  The CUDA source cannot be built here (no nvcc; UNBUILDABLE), and the reference
 * ships no vectors for it: PIN = analytic cases only (tests/test_postprocess_cpu.py:
 * axis-aligned rectangles, identical boxes, a square turned by 45 degrees, symmetry).
 * cos / sin / atan2 in double rounded to float (canonical form shared with the HIP kernel). */
typedef struct { float x, y; } op2;

static float ov_cross3(op2 p1, op2 p2, op2 p0) {
  return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y);
}

static int ov_crossing(op2 p1, op2 p0, op2 q1, op2 q0, op2 *ans) {
  if (!(fminf(p0.x, p1.x) <= fmaxf(q0.x, q1.x) && fminf(q0.x, q1.x) <= fmaxf(p0.x, p1.x) &&
        fminf(p0.y, p1.y) <= fmaxf(q0.y, q1.y) && fminf(q0.y, q1.y) <= fmaxf(p0.y, p1.y)))
    return 0;
  float s1 = ov_cross3(q0, p1, p0), s2 = ov_cross3(p1, q1, p0);
  float s3 = ov_cross3(p0, q1, q0), s4 = ov_cross3(q1, p1, q0);
  if (!(s1 * s2 > 0 && s3 * s4 > 0)) return 0;
  float s5 = ov_cross3(q1, p1, p0);
  if (fabsf(s5 - s1) > 1e-8f) {
    ans->x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
    ans->y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
  } else {
    float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
    float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
    float D = a0 * b1 - a1 * b0;
    ans->x = (b0 * c1 - b1 * c0) / D;
    ans->y = (a1 * c0 - a0 * c1) / D;
  }
  return 1;
}

static op2 ov_turn(float px, float py, float cx, float cy, float c, float s) {
  op2 r;
  r.x = (px - cx) * c + (py - cy) * s + cx;
  r.y = -(px - cx) * s + (py - cy) * c + cy;
  return r;
}

static void ov_corners(const float *b, op2 *c, float *cx, float *cy, float *ca, float *sa) {
  *cx = (b[0] + b[2]) / 2; *cy = (b[1] + b[3]) / 2;
  *ca = (float)cos((double)b[4]); *sa = (float)sin((double)b[4]);
  c[0] = ov_turn(b[0], b[1], *cx, *cy, *ca, *sa);
  c[1] = ov_turn(b[2], b[1], *cx, *cy, *ca, *sa);
  c[2] = ov_turn(b[2], b[3], *cx, *cy, *ca, *sa);
  c[3] = ov_turn(b[0], b[3], *cx, *cy, *ca, *sa);
  c[4] = c[0];
}

static int ov_inside(const float *b, float cx, float cy, float ca, float sa, op2 p) {
  const float M = 1e-5f;
  op2 q = ov_turn(p.x, p.y, cx, cy, ca, -sa); /* cos(-a), sin(-a) */
  return q.x > b[0] - M && q.x < b[2] + M && q.y > b[1] - M && q.y < b[3] + M;
}

static float ov_box_overlap(const float *a, const float *b) {
  op2 ca[5], cb[5], pts[16];
  float ang[16];
  float acx, acy, aco, asi, bcx, bcy, bco, bsi;
  ov_corners(a, ca, &acx, &acy, &aco, &asi);
  ov_corners(b, cb, &bcx, &bcy, &bco, &bsi);
  int cnt = 0;
  float sx = 0.f, sy = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      op2 hit;
      if (ov_crossing(ca[i + 1], ca[i], cb[j + 1], cb[j], &hit)) {
        sx = sx + hit.x; sy = sy + hit.y; pts[cnt++] = hit;
      }
    }
  for (int k = 0; k < 4; ++k) {
    if (ov_inside(a, acx, acy, aco, asi, cb[k])) { sx = sx + cb[k].x; sy = sy + cb[k].y; pts[cnt++] = cb[k]; }
    if (ov_inside(b, bcx, bcy, bco, bsi, ca[k])) { sx = sx + ca[k].x; sy = sy + ca[k].y; pts[cnt++] = ca[k]; }
  }
  if (cnt == 0) return 0.f;
  float mx = sx / cnt, my = sy / cnt;
  for (int i = 0; i < cnt; ++i) ang[i] = (float)atan2((double)(pts[i].y - my), (double)(pts[i].x - mx));
  /* the reference's bubble passes: swap neighbours while angle[i] > angle[i+1] */
  for (int j = 0; j < cnt - 1; ++j)
    for (int i = 0; i < cnt - j - 1; ++i)
      if (ang[i] > ang[i + 1]) {
        op2 t = pts[i]; pts[i] = pts[i + 1]; pts[i + 1] = t;
        float ta = ang[i]; ang[i] = ang[i + 1]; ang[i + 1] = ta;
      }
  float area = 0.f;
  for (int k = 0; k < cnt - 1; ++k) {
    float ax = pts[k].x - pts[0].x, ay = pts[k].y - pts[0].y;
    float bx = pts[k + 1].x - pts[0].x, by = pts[k + 1].y - pts[0].y;
    area += ax * by - ay * bx;
  }
  return (float)(fabsf(area) / 2.0);
}

int oracle_boxes_overlap_bev(int num_a, const float *boxes_a, int num_b, const float *boxes_b,
                             float *out) {
  for (int i = 0; i < num_a; ++i)
    for (int j = 0; j < num_b; ++j)
      out[(size_t)i * num_b + j] = ov_box_overlap(boxes_a + (size_t)i * 5, boxes_b + (size_t)j * 5);
  return 1;
}

/* ------------------------------------------------------------------ */
/* input side: batch assembly                         */
/* ------------------------------------------------------------------ */

/* The reference's per-sample pipeline on the points of one batch, in its order
 * (configs/Nesie/nesie-votenet-scannet-pretrain-010.py:151-196):
 *   GlobalAlignment   transforms_3d.py:465-488 -> points.rotate(R^T): p @ R^T, then + t
 *                     (base_points.py:173-177, 186-205); the height column is left alone
 *   IndoorPointSample :865-891   rows `choices`
 *   RandomFlip3D      :143-161   x = -x (horizontal), y = -y (vertical) (depth_points.py:28-33)
 *   GlobalRotScaleTrans :560-648 p @ [[c,-s,0],[s,c,0],[0,0,1]]^T, xyz and height * scale,
 *                     xyz + trans
 * xform (B,20) = R[9] t[3] | flip_x flip_y (-1/+1) | cos sin | scale | trans[3].
 * PIN: tests/golden/input_golden.pt = outputs of the reference's own transform classes
 * (loaded by path) on seeded scenes; agreement to float rounding (the reference multiplies
 * (N,3) by (3,3) through torch's matmul, whose summation order / FMA use is the BLAS's). */
int oracle_scene_assemble(int b, int n, long long pool_rows, const float *pool,
                          const float *height, const int *choices, const float *xform,
                          float *out) {
  for (int bi = 0; bi < b; ++bi) {
    const float *xf = xform + (size_t)bi * 20;
    for (int i = 0; i < n; ++i) {
      long long src = choices[(size_t)bi * n + i];
      if (src < 0) src = 0;
      if (src >= pool_rows) src = pool_rows - 1;
      const float x = pool[src * 3], y = pool[src * 3 + 1], z = pool[src * 3 + 2];
      float ax = ((x * xf[0] + y * xf[1]) + z * xf[2]) + xf[9];
      float ay = ((x * xf[3] + y * xf[4]) + z * xf[5]) + xf[10];
      float az = ((x * xf[6] + y * xf[7]) + z * xf[8]) + xf[11];
      if (xf[12] < 0.f) ax = -ax;
      if (xf[13] < 0.f) ay = -ay;
      const float c = xf[14], s = xf[15], sc = xf[16];
      const float rx = ax * c + ay * (-s);
      const float ry = ax * s + ay * c;
      float *o = out + ((size_t)bi * n + i) * 4;
      o[0] = rx * sc + xf[17];
      o[1] = ry * sc + xf[18];
      o[2] = az * sc + xf[19];
      o[3] = height[src] * sc;
    }
  }
  return 1;
}
