/* oracle/mcl3dl_oracle.c — TEST INFRASTRUCTURE ONLY. See mcl3dl_oracle.h for the parity status.
 *
 * A CPU restatement, in plain C, of the reference's per-particle LiDAR measurement update.
 * Every function cites the reference file:line (relative to /root/reference) it follows.
 * Arithmetic types (float vs double) and operation ORDER are the reference's; the build uses
 * -ffp-contract=off because the reference is compiled for baseline x86-64 (no FMA).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 */
#include "mcl3dl_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------------
 * Vec3 / Quat  (include/mcl_3dl/vec3.h, include/mcl_3dl/quat.h) — all float.
 * ---------------------------------------------------------------------------------------------- */
typedef struct
{
  float x, y, z;
} V3;
typedef struct
{
  float x, y, z, w;
} Q4;

static V3 v3(float x, float y, float z)
{
  V3 r = { x, y, z };
  return r;
}
static V3 v3_add(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }  /* vec3.h:101-104 */
static V3 v3_sub(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }  /* vec3.h:105-108 */
static V3 v3_scale(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }   /* vec3.h:113-116 */
static float v3_dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; } /* vec3.h:141-144 */
static float v3_norm(V3 a) { return sqrtf(v3_dot(a, a)); }                    /* vec3.h:156-159 */
static V3 v3_normalized(V3 a)                                                 /* vec3.h:160-163, operator/ :117-120 */
{
  const float n = v3_norm(a);
  return v3(a.x / n, a.y / n, a.z / n);
}

/* Hamilton product, quat.h:131-138 (term order is the reference's). */
static Q4 q_mul(Q4 a, Q4 q)
{
  Q4 r;
  r.x = a.w * q.x + a.x * q.w + a.y * q.z - a.z * q.y;
  r.y = a.w * q.y + a.y * q.w + a.z * q.x - a.x * q.z;
  r.z = a.w * q.z + a.z * q.w + a.x * q.y - a.y * q.x;
  r.w = a.w * q.w - a.x * q.x - a.y * q.y - a.z * q.z;
  return r;
}
/* Quat::operator*(Vec3), quat.h:139-143: (q * (v,0)) * conj(q). */
static V3 q_rot(Q4 q, V3 v)
{
  const Q4 qv = { v.x, v.y, v.z, 0.0f };
  const Q4 c = { -q.x, -q.y, -q.z, q.w }; /* conj, quat.h:183-186 */
  const Q4 r = q_mul(q_mul(q, qv), c);
  return v3(r.x, r.y, r.z);
}
/* Quat::normalized, quat.h:175-178 -> operator/(float) :148-151 -> operator*(1.0 / s): the reciprocal is formed
 * in double and narrowed to float when bound to operator*'s float parameter. */
static Q4 q_normalized(Q4 q)
{
  const float n = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w); /* quat.h:93-100 */
  const float s = (float)(1.0 / (double)n);
  Q4 r = { q.x * s, q.y * s, q.z * s, q.w * s };
  return r;
}

/* ------------------------------------------------------------------------------------------------
 * State
 * ---------------------------------------------------------------------------------------------- */
typedef struct
{
  /* RaycastUsingDDA members, include/mcl_3dl/raycasts/raycast_using_dda.h:272-298 */
  double min_dist_thr_sq, dda_grid_size, ray_angle_half, hit_tolerance;
  int map_size[3];
  float min_p[3], max_p[3];
  uint8_t* exists; /* point_exists_ */
  size_t total;
  /* points_ (unordered_map<size_t, vector<const Point*>>) as sorted keys + CSR, insertion order kept */
  size_t n_occ;
  size_t* occ_key;
  uint32_t* occ_start;
  uint32_t* occ_pts;
  int built;
  uint64_t stamp;
  /* per-ray state */
  int begin_index[3], end_index[3], current_index[3], step[3];
  V3 ray_begin, ray_dir;
  int max_movement, pos;
  float t_max[3], t_delta[3], initial_edges[3];
  /* statistics (oracle-only) */
  double n_steps, n_occupied, n_tested;
} Dda;

typedef struct
{
  float map_grid_min, map_grid_max, hit_tolerance;
  V3 pos, inc;
  int length, count;
} KdRay;

typedef struct
{
  /* ChunkedKdtree, include/mcl_3dl/chunked_kdtree.h:91-102,274-280 */
  float pos_to_chunk, chunk_length, max_search_radius;
  /* map */
  size_t n;
  float* xyz;
  uint32_t* label;
  uint64_t stamp;
  int has_weight;
  float weight[3];
  int32_t* chunk; /* 3 per point: getChunkId */
  int8_t* bound;  /* 3 per point: x_bound,y_bound,z_bound of setInputCloud */
  /* exact-NN grid over the rescaled coordinates (ours; results do not depend on it) */
  float* sxyz;
  float g_origin[3], g_cell;
  int g_dim[3];
  uint32_t* g_start;
  uint32_t* g_ids;
  /* LidarMeasurementModelLikelihoodParameters (parameters.h:64-89) */
  float match_dist_min, match_dist_flat, match_weight;
  /* LidarMeasurementModelBeamParameters (parameters.h:91-132) + derived (beam.cpp:58-80) */
  float map_grid[3], dda_grid_size, ray_angle_half, hit_range, beam_likelihood_min, ang_total_ref;
  uint32_t beam_num_points, filter_label_max;
  int short_only, use_dda;
  float hit_range_sq, beam_likelihood, sin_total_ref, search_range;
  int dda_params_epoch;
} Oracle;

static double now_sec(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ------------------------------------------------------------------------------------------------
 * ChunkedKdtree + pcl::KdTreeFLANN::radiusSearch(max_nn = 1)
 * ---------------------------------------------------------------------------------------------- */
/* getChunkId, chunked_kdtree.h:258-263: floor(p * pos_to_chunk) in float. */
static void chunk_id(const Oracle* o, const float* p, int32_t* c)
{
  c[0] = (int32_t)floorf(p[0] * o->pos_to_chunk);
  c[1] = (int32_t)floorf(p[1] * o->pos_to_chunk);
  c[2] = (int32_t)floorf(p[2] * o->pos_to_chunk);
}

/* PointRepresentation::vectorize (third-party PCL; restated in oracle/shims/pcl/kdtree/kdtree.h):
 * out[i] = in[i] * alpha[i] when rescale values are set. */
static void vectorize(const Oracle* o, const float* p, float* out)
{
  if (o->has_weight)
  {
    out[0] = p[0] * o->weight[0];
    out[1] = p[1] * o->weight[1];
    out[2] = p[2] * o->weight[2];
  }
  else
  {
    out[0] = p[0];
    out[1] = p[1];
    out[2] = p[2];
  }
}

/* flann::L2_Simple<float> (third-party; published algorithm): result += diff*diff per dimension, float. */
static float l2_simple(const float* a, const float* b)
{
  float result = 0.0f;
  for (int i = 0; i < 3; ++i)
  {
    const float diff = a[i] - b[i];
    result += diff * diff;
  }
  return result;
}

static void free_map(Oracle* o)
{
  free(o->xyz);
  free(o->label);
  free(o->chunk);
  free(o->bound);
  free(o->sxyz);
  free(o->g_start);
  free(o->g_ids);
  o->xyz = o->sxyz = NULL;
  o->label = NULL;
  o->chunk = NULL;
  o->bound = NULL;
  o->g_start = o->g_ids = NULL;
  o->n = 0;
}

static uint32_t grid_cell_of(const Oracle* o, const float* v)
{
  int c[3];
  for (int a = 0; a < 3; ++a)
  {
    c[a] = (int)floorf((v[a] - o->g_origin[a]) / o->g_cell);
    if (c[a] < 0)
      c[a] = 0;
    if (c[a] > o->g_dim[a] - 1)
      c[a] = o->g_dim[a] - 1;
  }
  return (uint32_t)(((size_t)c[2] * o->g_dim[1] + c[1]) * o->g_dim[0] + c[0]);
}

/* ChunkedKdtree::setInputCloud, chunked_kdtree.h:124-216.  The reference copies every point into the
 * cloud of its own chunk and of up to 7 neighbour chunks when it lies within max_search_radius_ of a
 * chunk face; we keep, per point, its chunk id and the three *_bound flags, from which membership of a
 * point in any chunk's cloud follows (see in_chunk_cloud). */
void orc_set_map(void* h, const float* xyz, const uint32_t* label, size_t n, uint64_t stamp, const float* dist_weight,
                 float epsilon)
{
  Oracle* o = (Oracle*)h;
  (void)epsilon; /* FLANN eps: accepted, ignored (exact search) — "parity unpinned", see header */
  free_map(o);
  o->n = n;
  o->stamp = stamp;
  o->xyz = (float*)malloc(sizeof(float) * 3 * (n ? n : 1));
  o->label = (uint32_t*)malloc(sizeof(uint32_t) * (n ? n : 1));
  o->chunk = (int32_t*)malloc(sizeof(int32_t) * 3 * (n ? n : 1));
  o->bound = (int8_t*)malloc(3 * (n ? n : 1));
  o->sxyz = (float*)malloc(sizeof(float) * 3 * (n ? n : 1));
  memcpy(o->xyz, xyz, sizeof(float) * 3 * n);
  for (size_t i = 0; i < n; ++i)
    o->label[i] = label ? label[i] : 0;
  if (dist_weight)
  {
    o->has_weight = 1;
    memcpy(o->weight, dist_weight, sizeof(float) * 3);
  }
  float mn[3] = { 0, 0, 0 }, mx[3] = { 0, 0, 0 };
  for (size_t i = 0; i < n; ++i)
  {
    const float* p = &o->xyz[3 * i];
    int32_t* c = &o->chunk[3 * i];
    chunk_id(o, p, c);
    for (int a = 0; a < 3; ++a)
    {
      const float in_chunk = p[a] - c[a] * o->chunk_length; /* chunked_kdtree.h:139,146,153 */
      int8_t b = 0;
      if (in_chunk < o->max_search_radius)
        b = -1;
      else if (in_chunk > o->chunk_length - o->max_search_radius)
        b = 1;
      o->bound[3 * i + a] = b;
    }
    vectorize(o, p, &o->sxyz[3 * i]);
    for (int a = 0; a < 3; ++a)
    {
      const float v = o->sxyz[3 * i + a];
      if (i == 0 || v < mn[a])
        mn[a] = v;
      if (i == 0 || v > mx[a])
        mx[a] = v;
    }
  }
  o->g_cell = 0.25f;
  for (;;)
  {
    double total = 1;
    for (int a = 0; a < 3; ++a)
    {
      o->g_dim[a] = (int)floorf((mx[a] - mn[a]) / o->g_cell) + 1;
      total *= o->g_dim[a];
    }
    if (total <= 256.0 * 1024 * 1024)
      break;
    o->g_cell *= 2;
  }
  for (int a = 0; a < 3; ++a)
    o->g_origin[a] = mn[a];
  const size_t ncell = (size_t)o->g_dim[0] * o->g_dim[1] * o->g_dim[2];
  o->g_start = (uint32_t*)calloc(ncell + 1, sizeof(uint32_t));
  o->g_ids = (uint32_t*)malloc(sizeof(uint32_t) * (n ? n : 1));
  uint32_t* cell_of = (uint32_t*)malloc(sizeof(uint32_t) * (n ? n : 1));
  for (size_t i = 0; i < n; ++i)
  {
    cell_of[i] = grid_cell_of(o, &o->sxyz[3 * i]);
    ++o->g_start[cell_of[i] + 1];
  }
  for (size_t c = 0; c < ncell; ++c)
    o->g_start[c + 1] += o->g_start[c];
  uint32_t* fill = (uint32_t*)malloc(sizeof(uint32_t) * (ncell ? ncell : 1));
  memcpy(fill, o->g_start, sizeof(uint32_t) * ncell);
  for (size_t i = 0; i < n; ++i)
    o->g_ids[fill[cell_of[i]]++] = (uint32_t)i;
  free(fill);
  free(cell_of);
}

/* Is map point i in the cloud that setInputCloud built for chunk qc?  chunked_kdtree.h:136-201:
 * the point goes to its own chunk and to chunk + (bx*sx, by*sy, bz*sz) for every non-empty subset of the
 * axes whose *_bound is non-zero. */
static int in_chunk_cloud(const Oracle* o, size_t i, const int32_t* qc)
{
  for (int a = 0; a < 3; ++a)
  {
    const int d = qc[a] - o->chunk[3 * i + a];
    if (d != 0 && d != o->bound[3 * i + a])
      return 0;
  }
  return 1;
}

/* ChunkedKdtree::radiusSearch(p, radius, id, dist_sq, 1), chunked_kdtree.h:217-237, on top of
 * pcl::KdTreeFLANN::radiusSearch: nearest point of the query's chunk cloud with d2 < (float)(radius*radius). */
static int radius_search1(const Oracle* o, const float* p, float radius, int* id, float* sqdist)
{
  if (o->n == 0)
    return 0;
  int32_t qc[3];
  chunk_id(o, p, qc);
  float q[3];
  vectorize(o, p, q);
  const float r2 = (float)((double)radius * (double)radius);
  int lo[3], hi[3];
  for (int a = 0; a < 3; ++a)
  {
    const double l = floor(((double)q[a] - radius - o->g_origin[a]) / o->g_cell) - 1;
    const double u = floor(((double)q[a] + radius - o->g_origin[a]) / o->g_cell) + 1;
    if (u < 0 || l > o->g_dim[a] - 1)
      return 0;
    lo[a] = (int)(l < 0 ? 0 : l);
    hi[a] = (int)(u > o->g_dim[a] - 1 ? o->g_dim[a] - 1 : u);
  }
  float best = r2;
  long best_id = -1;
  for (int z = lo[2]; z <= hi[2]; ++z)
    for (int y = lo[1]; y <= hi[1]; ++y)
    {
      const size_t row = ((size_t)z * o->g_dim[1] + y) * o->g_dim[0];
      const uint32_t b = o->g_start[row + lo[0]], e = o->g_start[row + hi[0] + 1];
      for (uint32_t k = b; k < e; ++k)
      {
        const uint32_t i = o->g_ids[k];
        if (!in_chunk_cloud(o, i, qc))
          continue;
        const float d2 = l2_simple(q, &o->sxyz[3 * i]);
        if (d2 < best || (d2 == best && best_id >= 0 && (long)i < best_id))
        {
          best = d2;
          best_id = (long)i;
        }
      }
    }
  if (best_id < 0)
    return 0;
  *id = (int)best_id;
  *sqdist = best;
  return 1;
}

void orc_radius_search(void* h, const float* q_xyz, size_t n, float radius, int* found, int* id, float* sqdist)
{
  const Oracle* o = (const Oracle*)h;
  for (size_t i = 0; i < n; ++i)
  {
    int k = -1;
    float d = -1.f;
    found[i] = radius_search1(o, &q_xyz[3 * i], radius, &k, &d);
    id[i] = found[i] ? k : -1;
    sqdist[i] = found[i] ? d : -1.f;
  }
}

/* ------------------------------------------------------------------------------------------------
 * State6DOF::transform, include/mcl_3dl/state_6dof.h:214-225
 * ---------------------------------------------------------------------------------------------- */
static void transform_points(const float* pose7, const float* in, size_t n, float* out)
{
  const V3 pos = v3(pose7[0], pose7[1], pose7[2]);
  const Q4 rot = { pose7[3], pose7[4], pose7[5], pose7[6] };
  const Q4 r = q_normalized(rot);
  for (size_t i = 0; i < n; ++i)
  {
    const V3 t = v3_add(q_rot(r, v3(in[3 * i], in[3 * i + 1], in[3 * i + 2])), pos);
    out[3 * i + 0] = t.x;
    out[3 * i + 1] = t.y;
    out[3 * i + 2] = t.z;
  }
}

void orc_transform(const float* pose7, const float* xyz_in, size_t n, float* xyz_out)
{
  transform_points(pose7, xyz_in, n, xyz_out);
}

void orc_quat_rotate(const float* q4, const float* v3in, float* out3)
{
  const Q4 q = { q4[0], q4[1], q4[2], q4[3] };
  const V3 r = q_rot(q, v3(v3in[0], v3in[1], v3in[2]));
  out3[0] = r.x;
  out3[1] = r.y;
  out3[2] = r.z;
}

/* ------------------------------------------------------------------------------------------------
 * LidarMeasurementModelLikelihood::measure, src/lidar_measurement_model_likelihood.cpp:105-139
 * ---------------------------------------------------------------------------------------------- */
static void likelihood_measure1(const Oracle* o, const float* pose7, const float* scan_xyz, size_t n_s, float* tmp,
                                float* lik, float* quality)
{
  if (n_s == 0) /* :111-114 (null or empty cloud) */
  {
    *lik = 1.0f;
    *quality = 0.0f;
    return;
  }
  transform_points(pose7, scan_xyz, n_s, tmp); /* :121-122 */
  float score_like = 0;
  size_t num = 0;
  for (size_t i = 0; i < n_s; ++i)
  {
    int id;
    float sqdist;
    if (radius_search1(o, &tmp[3 * i], o->match_dist_min, &id, &sqdist)) /* :126 */
    {
      const float s = sqrtf(sqdist);
      const float dist = o->match_dist_min - (s > o->match_dist_flat ? s : o->match_dist_flat); /* :128 std::max(a,b): a<b?b:a */
      if (dist < 0.0)
        continue;
      score_like += dist * o->match_weight; /* :132 */
      num++;
    }
  }
  *lik = score_like;
  *quality = (float)num / n_s; /* :136: float / size_t -> float */
}

double orc_likelihood_measure(void* h, const float* poses, size_t n_p, const float* scan_xyz,
                              const uint32_t* scan_label, size_t n_s, float* out_lik, float* out_quality, int threads)
{
  const Oracle* o = (const Oracle*)h;
  (void)scan_label;
  const int nt = threads > 1 ? threads : 1;
  const double t0 = now_sec();
#pragma omp parallel num_threads(nt)
  {
    float* tmp = (float*)malloc(sizeof(float) * 3 * (n_s ? n_s : 1));
#pragma omp for schedule(dynamic, 1)
    for (long i = 0; i < (long)n_p; ++i)
      likelihood_measure1(o, &poses[7 * i], scan_xyz, n_s, tmp, &out_lik[i], &out_quality[i]);
    free(tmp);
  }
  return now_sec() - t0;
}

/* ------------------------------------------------------------------------------------------------
 * RaycastUsingDDA, include/mcl_3dl/raycasts/raycast_using_dda.h
 * ---------------------------------------------------------------------------------------------- */
static void dda_free(Dda* d)
{
  free(d->exists);
  free(d->occ_key);
  free(d->occ_start);
  free(d->occ_pts);
  d->exists = NULL;
  d->occ_key = NULL;
  d->occ_start = d->occ_pts = NULL;
  d->built = 0;
  d->n_occ = 0;
}

/* ctor, raycast_using_dda.h:56-64.  min_dist_thr_sq_ uses map_grid_size_y twice (reference quirk, kept). */
static void dda_init(Dda* d, double gx, double gy, double gz, double dda_grid_size, double ray_angle_half,
                     double hit_tolerance)
{
  memset(d, 0, sizeof(*d));
  (void)gz;
  d->min_dist_thr_sq = gx * gx + gy * gy + gy * gy;
  d->dda_grid_size = dda_grid_size;
  d->ray_angle_half = ray_angle_half;
  d->hit_tolerance = hit_tolerance;
}

/* toIndex(point), raycast_using_dda.h:205-217: float difference, double division, truncation toward zero. */
static void dda_to_index(const Dda* d, const float* p, int* idx)
{
  idx[0] = (int)((p[0] - d->min_p[0]) / d->dda_grid_size);
  idx[1] = (int)((p[1] - d->min_p[1]) / d->dda_grid_size);
  idx[2] = (int)((p[2] - d->min_p[2]) / d->dda_grid_size);
}
/* getArrayIndex, :225-228 */
static size_t dda_array_index(const Dda* d, const int* c)
{
  return (size_t)(c[0] + c[1] * d->map_size[0] + c[2] * (d->map_size[0] * d->map_size[1]));
}

typedef struct
{
  size_t key;
  uint32_t idx;
} KeyIdx;
static int cmp_keyidx(const void* a, const void* b)
{
  const KeyIdx* x = (const KeyIdx*)a;
  const KeyIdx* y = (const KeyIdx*)b;
  if (x->key != y->key)
    return x->key < y->key ? -1 : 1;
  return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0);
}

/* updatePointCloud, raycast_using_dda.h:162-190 (+ setExists :230-235). */
static void dda_update(Dda* d, const Oracle* o)
{
  if (d->built && d->stamp == o->stamp && d->n_occ != 0) /* :168 */
    return;
  dda_free(d);
  d->stamp = o->stamp;
  /* pcl::getMinMax3D */
  for (int a = 0; a < 3; ++a)
  {
    d->min_p[a] = FLT_MAX;
    d->max_p[a] = -FLT_MAX;
  }
  for (size_t i = 0; i < o->n; ++i)
    for (int a = 0; a < 3; ++a)
    {
      const float v = o->xyz[3 * i + a];
      if (v < d->min_p[a])
        d->min_p[a] = v;
      if (v > d->max_p[a])
        d->max_p[a] = v;
    }
  int point_total = 1;
  for (int a = 0; a < 3; ++a)
  {
    d->map_size[a] = (int)((size_t)((d->max_p[a] - d->min_p[a]) / d->dda_grid_size) + 1); /* :178 */
    point_total *= d->map_size[a];
  }
  d->total = (size_t)point_total;
  d->exists = (uint8_t*)calloc(d->total ? d->total : 1, 1);
  KeyIdx* ki = (KeyIdx*)malloc(sizeof(KeyIdx) * (o->n ? o->n : 1));
  for (size_t i = 0; i < o->n; ++i)
  {
    int idx[3];
    dda_to_index(d, &o->xyz[3 * i], idx);
    const size_t k = dda_array_index(d, idx);
    d->exists[k] = 1;
    ki[i].key = k;
    ki[i].idx = (uint32_t)i;
  }
  qsort(ki, o->n, sizeof(KeyIdx), cmp_keyidx); /* (key, insertion index): insertion order kept inside a voxel */
  d->occ_key = (size_t*)malloc(sizeof(size_t) * (o->n ? o->n : 1));
  d->occ_start = (uint32_t*)malloc(sizeof(uint32_t) * (o->n + 1));
  d->occ_pts = (uint32_t*)malloc(sizeof(uint32_t) * (o->n ? o->n : 1));
  d->n_occ = 0;
  for (size_t i = 0; i < o->n; ++i)
  {
    if (i == 0 || ki[i].key != ki[i - 1].key)
    {
      d->occ_key[d->n_occ] = ki[i].key;
      d->occ_start[d->n_occ] = (uint32_t)i;
      ++d->n_occ;
    }
    d->occ_pts[i] = ki[i].idx;
  }
  d->occ_start[d->n_occ] = (uint32_t)o->n;
  free(ki);
  d->built = 1;
}

/* isPointWithinMap, :260-270 */
static int dda_within_map(const Dda* d, V3 p)
{
  const float v[3] = { p.x, p.y, p.z };
  for (int i = 0; i < 3; ++i)
    if ((v[i] < d->min_p[i]) || (d->max_p[i] < v[i]))
      return 0;
  return 1;
}

/* setRay, :66-104 */
static void dda_set_ray(Dda* d, const Oracle* o, V3 ray_begin, V3 ray_end_org)
{
  dda_update(d, o);
  if (!dda_within_map(d, ray_begin))
  {
    d->max_movement = 0;
    d->pos = 0;
    return;
  }
  d->ray_begin = ray_begin;
  d->ray_dir = v3_normalized(v3_sub(ray_end_org, ray_begin));
  /* Vec3 * double: operator*(const float s) — hit_tolerance_ narrows to float */
  const V3 ray_end = v3_add(ray_end_org, v3_scale(d->ray_dir, (float)d->hit_tolerance));
  const float b[3] = { ray_begin.x, ray_begin.y, ray_begin.z };
  const float e[3] = { ray_end.x, ray_end.y, ray_end.z };
  const float dir[3] = { d->ray_dir.x, d->ray_dir.y, d->ray_dir.z };
  dda_to_index(d, b, d->begin_index);
  dda_to_index(d, e, d->end_index);
  int dist[3];
  for (int i = 0; i < 3; ++i)
    dist[i] = d->end_index[i] - d->begin_index[i];
  d->pos = 0;
  d->max_movement = abs(dist[0]) + abs(dist[1]) + abs(dist[2]);
  for (int i = 0; i < 3; ++i)
    d->step[i] = (dist[i] < 0) ? -1 : 1;
  for (int i = 0; i < 3; ++i)
    d->current_index[i] = d->begin_index[i];
  for (int i = 0; i < 3; ++i)
  {
    if (dist[i] == 0)
    {
      d->initial_edges[i] = (float)INFINITY;
      d->t_delta[i] = (float)INFINITY;
    }
    else
    {
      const double nearest = (dir[i] < 0) ? d->begin_index[i] * d->dda_grid_size + d->min_p[i] :
                                            (d->begin_index[i] + 1) * d->dda_grid_size + d->min_p[i];
      d->initial_edges[i] = (float)fabs((nearest - b[i]) / dir[i]); /* double expression stored in a float Vec3 */
      d->t_delta[i] = (float)fabs(d->dda_grid_size / dir[i]);
    }
  }
  for (int i = 0; i < 3; ++i)
    d->t_max[i] = d->initial_edges[i];
}

/* incrementIndex, :192-203 */
static int dda_increment(Dda* d, int i)
{
  d->current_index[i] += d->step[i];
  d->t_max[i] = d->initial_edges[i] + d->t_delta[i] * abs(d->current_index[i] - d->begin_index[i]);
  if (d->current_index[i] < 0 || d->map_size[i] <= d->current_index[i])
  {
    d->pos = d->max_movement;
    return 0;
  }
  return 1;
}

/* hasIntersection, :237-258.  Returns map point index or -1. */
static long dda_has_intersection(Dda* d, const Oracle* o)
{
  const size_t array_index = dda_array_index(d, d->current_index);
  if (!d->exists[array_index])
    return -1;
  d->n_occupied += 1;
  size_t lo = 0, hi = d->n_occ;
  while (lo < hi)
  {
    const size_t mid = (lo + hi) / 2;
    if (d->occ_key[mid] < array_index)
      lo = mid + 1;
    else
      hi = mid;
  }
  for (uint32_t k = d->occ_start[lo]; k < d->occ_start[lo + 1]; ++k)
  {
    const uint32_t pi = d->occ_pts[k];
    const float* t = &o->xyz[3 * pi];
    d->n_tested += 1;
    const V3 target_rel = v3(t[0] - d->ray_begin.x, t[1] - d->ray_begin.y, t[2] - d->ray_begin.z);
    const double dist_to_perpendicular_foot = fabsf(v3_dot(target_rel, d->ray_dir));
    const double a = d->ray_angle_half * dist_to_perpendicular_foot;
    const double a2 = a * a; /* std::pow(x, 2) */
    const double dist_threshold_sq = a2 < d->min_dist_thr_sq ? d->min_dist_thr_sq : a2; /* std::max(a2, thr) */
    const double dist_sq = v3_dot(target_rel, target_rel) - dist_to_perpendicular_foot * dist_to_perpendicular_foot;
    if (dist_sq < dist_threshold_sq)
      return (long)pi;
  }
  return -1;
}

/* getNextCastResult, :106-159.  returns 0 when the ray is exhausted; *collided_point = map index or -1. */
static int dda_next(Dda* d, const Oracle* o, V3* pos, long* collided_point)
{
  ++d->pos;
  if (d->pos >= d->max_movement)
    return 0;
  int axis;
  if (d->t_max[0] < d->t_max[1])
    axis = (d->t_max[0] < d->t_max[2]) ? 0 : 2;
  else
    axis = (d->t_max[1] < d->t_max[2]) ? 1 : 2;
  if (!dda_increment(d, axis))
    return 0;
  d->n_steps += 1;
  *collided_point = dda_has_intersection(d, o);
  /* fromIndex, :219-223: double expression narrowed to the float Vec3 */
  pos->x = (float)((d->current_index[0] + 0.5) * d->dda_grid_size + d->min_p[0]);
  pos->y = (float)((d->current_index[1] + 0.5) * d->dda_grid_size + d->min_p[1]);
  pos->z = (float)((d->current_index[2] + 0.5) * d->dda_grid_size + d->min_p[2]);
  return 1;
}

/* ------------------------------------------------------------------------------------------------
 * RaycastUsingKDTree, include/mcl_3dl/raycasts/raycast_using_kdtree.h:58-109 (oracle only; not on the GPU path)
 * ---------------------------------------------------------------------------------------------- */
static void kdray_set(KdRay* k, V3 ray_begin, V3 ray_end)
{
  const V3 diff = v3_sub(ray_end, ray_begin);
  k->length = (int)floorf((v3_norm(diff) + k->hit_tolerance) / k->map_grid_min); /* :61 */
  k->inc = v3_scale(v3_normalized(diff), k->map_grid_min);                       /* :62 */
  k->count = 1;
  k->pos = v3_add(ray_begin, k->inc);
}

static int kdray_next(KdRay* k, const Oracle* o, V3* pos, int* collision, float* sin_angle, long* point)
{
  if (k->count >= k->length)
    return 0;
  *collision = 0;
  float sin_ang = 0.0f;
  *point = -1;
  const float c[3] = { k->pos.x, k->pos.y, k->pos.z };
  int id;
  float sqdist;
  if (radius_search1(o, c, (float)(sqrt(2.0) * k->map_grid_max / 2.0), &id, &sqdist)) /* :83 */
  {
    *collision = 1;
    *point = id;
    const float d0 = sqrtf(sqdist);
    const V3 pos_prev = v3_sub(k->pos, v3_scale(k->inc, 2.0f));
    const float cp[3] = { pos_prev.x, pos_prev.y, pos_prev.z };
    if (radius_search1(o, cp, (float)(k->map_grid_min * 2 + sqrt(2.0) * k->map_grid_max / 2.0), &id, &sqdist)) /* :94 */
    {
      const float d1 = sqrtf(sqdist);
      sin_ang = (float)(fabs(d1 - d0) / (k->map_grid_min * 2.0)); /* :97 */
    }
    else
    {
      sin_ang = 1.0f;
    }
  }
  *pos = k->pos;
  *sin_angle = sin_ang;
  ++k->count;
  k->pos = v3_add(k->pos, k->inc);
  return 1;
}

/* ------------------------------------------------------------------------------------------------
 * LidarMeasurementModelBeam, src/lidar_measurement_model_beam.cpp
 * ---------------------------------------------------------------------------------------------- */
/* refreshParameters, beam.cpp:58-80 */
static void beam_refresh(Oracle* o)
{
  float m = o->map_grid[0];
  if (o->map_grid[1] > m)
    m = o->map_grid[1];
  if (o->map_grid[2] > m)
    m = o->map_grid[2];
  o->search_range = m * 4;
  o->hit_range_sq = (float)pow((double)o->hit_range, 2);                                            /* :65 */
  o->beam_likelihood = (float)pow((double)o->beam_likelihood_min, 1.0 / (float)o->beam_num_points); /* :66 */
  o->sin_total_ref = sinf(o->ang_total_ref);                                                        /* :67 */
  o->dda_params_epoch++;
}

typedef struct
{
  Dda dda;
  KdRay kd;
  int epoch;
  int init;
} Caster;

static void caster_prepare(Caster* c, const Oracle* o)
{
  if (c->init && c->epoch == o->dda_params_epoch)
    return;
  if (c->init)
    dda_free(&c->dda);
  dda_init(&c->dda, o->map_grid[0], o->map_grid[1], o->map_grid[2], o->dda_grid_size, o->ray_angle_half,
           o->hit_range); /* beam.cpp:71-73: doubles receive the float params */
  float mn = o->map_grid[0], mx = o->map_grid[0];
  for (int a = 1; a < 3; ++a)
  {
    if (o->map_grid[a] < mn)
      mn = o->map_grid[a];
    if (o->map_grid[a] > mx)
      mx = o->map_grid[a];
  }
  c->kd.map_grid_min = mn;
  c->kd.map_grid_max = mx;
  c->kd.hit_tolerance = o->hit_range;
  c->epoch = o->dda_params_epoch;
  c->init = 1;
}

/* getBeamStatus, beam.cpp:157-192.  0 SHORT, 1 HIT, 2 LONG, 3 TOTAL_REFLECTION. */
static int beam_status1(Caster* c, const Oracle* o, V3 lidar_pos, V3 scan_pos, long* hit_point)
{
  *hit_point = -1;
  if (o->use_dda)
    dda_set_ray(&c->dda, o, lidar_pos, scan_pos);
  else
    kdray_set(&c->kd, lidar_pos, scan_pos);
  for (;;)
  {
    V3 pos;
    long point = -1;
    int collision = 0;
    float sin_angle = 1.0f;
    if (o->use_dda)
    {
      if (!dda_next(&c->dda, o, &pos, &point))
        break;
      collision = point >= 0;
      sin_angle = 1.0f; /* raycast_using_dda.h:152,156 */
    }
    else
    {
      if (!kdray_next(&c->kd, o, &pos, &collision, &sin_angle, &point))
        break;
    }
    if (!collision)
      continue;
    if (o->label[point] > o->filter_label_max) /* :168 */
      continue;
    *hit_point = point;
    if (sin_angle > o->sin_total_ref) /* :171 */
    {
      const float* m = &o->xyz[3 * point];
      const double dx = (double)(scan_pos.x - m[0]), dy = (double)(scan_pos.y - m[1]),
                   dz = (double)(scan_pos.z - m[2]);
      const float distance_from_point_sq = (float)(dx * dx + dy * dy + dz * dz); /* :173-175 std::pow(float,2)->double */
      if (distance_from_point_sq < o->hit_range_sq)
        return 1;
      return 0;
    }
    return 3;
  }
  return 2;
}

/* measure, beam.cpp:124-155 */
static void beam_measure1(Caster* c, const Oracle* o, const float* pose7, const float* scan_xyz,
                          const uint32_t* scan_label, size_t n_b, const float* origins_xyz, float* tmp, float* lik,
                          float* quality)
{
  if (n_b == 0) /* :130-133 */
  {
    *lik = 1.0f;
    *quality = 0.0f;
    return;
  }
  float score_beam = 1.0f;
  transform_points(pose7, scan_xyz, n_b, tmp);
  const V3 s_pos = v3(pose7[0], pose7[1], pose7[2]);
  const Q4 s_rot = { pose7[3], pose7[4], pose7[5], pose7[6] }; /* NOT normalised here, :145 */
  for (size_t i = 0; i < n_b; ++i)
  {
    const int beam_header_id = (int)scan_label[i];
    const float* og = &origins_xyz[3 * beam_header_id];
    const V3 lidar_pos = v3_add(s_pos, q_rot(s_rot, v3(og[0], og[1], og[2])));
    long hp;
    const int status = beam_status1(c, o, lidar_pos, v3(tmp[3 * i], tmp[3 * i + 1], tmp[3 * i + 2]), &hp);
    if ((status == 0) || (!o->short_only && (status == 2)))
      score_beam *= o->beam_likelihood;
  }
  if (score_beam < o->beam_likelihood_min)
    score_beam = o->beam_likelihood_min;
  *lik = score_beam;
  *quality = 1.0f;
}

double orc_beam_measure(void* h, const float* poses, size_t n_p, const float* scan_xyz, const uint32_t* scan_label,
                        size_t n_b, const float* origins_xyz, size_t n_o, float* out_lik, float* out_quality,
                        int threads)
{
  const Oracle* o = (const Oracle*)h;
  (void)n_o;
  const int nt = threads > 1 ? threads : 1;
  double t_total = 0;
#pragma omp parallel num_threads(nt)
  {
    Caster c;
    memset(&c, 0, sizeof(c));
    caster_prepare(&c, o);
    if (o->use_dda && n_b)
      dda_update(&c.dda, o); /* grid build outside the timed region (reference builds once per map stamp) */
    float* tmp = (float*)malloc(sizeof(float) * 3 * (n_b ? n_b : 1));
#pragma omp barrier
#pragma omp master
    t_total = now_sec();
#pragma omp for schedule(dynamic, 1)
    for (long i = 0; i < (long)n_p; ++i)
      beam_measure1(&c, o, &poses[7 * i], scan_xyz, scan_label, n_b, origins_xyz, tmp, &out_lik[i], &out_quality[i]);
#pragma omp master
    t_total = now_sec() - t_total;
    free(tmp);
    dda_free(&c.dda);
  }
  return t_total;
}

void orc_beam_status(void* h, const float* begin_xyz, const float* end_xyz, size_t n, int* status, int* hit_index)
{
  const Oracle* o = (const Oracle*)h;
  Caster c;
  memset(&c, 0, sizeof(c));
  caster_prepare(&c, o);
  for (size_t i = 0; i < n; ++i)
  {
    long hp;
    status[i] = beam_status1(&c, o, v3(begin_xyz[3 * i], begin_xyz[3 * i + 1], begin_xyz[3 * i + 2]),
                             v3(end_xyz[3 * i], end_xyz[3 * i + 1], end_xyz[3 * i + 2]), &hp);
    if (hit_index)
      hit_index[i] = (status[i] != 2) ? (int)hp : -1;
  }
  dda_free(&c.dda);
}

int orc_dda_waypoints(void* h, double map_grid_x, double map_grid_y, double map_grid_z, double dda_grid_size,
                      double ray_angle_half, double hit_tolerance, const float* begin3, const float* end3,
                      float* out_xyz, int max_out, int* collided, int* hit_index, int stop_at_collision)
{
  const Oracle* o = (const Oracle*)h;
  Dda d;
  dda_init(&d, map_grid_x, map_grid_y, map_grid_z, dda_grid_size, ray_angle_half, hit_tolerance);
  dda_set_ray(&d, o, v3(begin3[0], begin3[1], begin3[2]), v3(end3[0], end3[1], end3[2]));
  int n = 0;
  *collided = 0;
  if (hit_index)
    *hit_index = -1;
  V3 pos;
  long point;
  while (dda_next(&d, o, &pos, &point))
  {
    if (n < max_out)
    {
      out_xyz[3 * n + 0] = pos.x;
      out_xyz[3 * n + 1] = pos.y;
      out_xyz[3 * n + 2] = pos.z;
    }
    ++n;
    if (point >= 0)
    {
      if (!*collided && hit_index)
        *hit_index = (int)point;
      *collided = 1;
      if (stop_at_collision)
        break;
    }
  }
  dda_free(&d);
  return n;
}

/* ------------------------------------------------------------------------------------------------
 * pf::ParticleFilter::measure, include/mcl_3dl/pf.h:252-279
 * ---------------------------------------------------------------------------------------------- */
static int pf_measure_core(float* w, const float* likelihood, size_t n, float* entropy)
{
  float* prev = (float*)malloc(sizeof(float) * (n ? n : 1)); /* particles_prev, :254 */
  memcpy(prev, w, sizeof(float) * n);
  float sum = 0;
  for (size_t i = 0; i < n; ++i)
  {
    w[i] *= likelihood[i];
    sum += w[i];
  }
  int restored = 0;
  if (sum > 0.0)
  {
    float e = 0;
    for (size_t i = 0; i < n; ++i)
    {
      w[i] /= sum;
      if (w[i] > 0)
        e += w[i] * logf(w[i]); /* std::log(float) */
    }
    e *= -1;
    *entropy = e;
  }
  else
  {
    memcpy(w, prev, sizeof(float) * n);
    *entropy = NAN; /* entropy_ is left untouched by the reference */
    restored = 1;
  }
  free(prev);
  return restored;
}

int orc_pf_measure(float* weight_inout, const float* likelihood, size_t n, float* entropy)
{
  return pf_measure_core(weight_inout, likelihood, n, entropy);
}

/* The node's measure lambda + pf::measure, src/mcl_3dl.cpp:398-426.  std::map iteration order puts "beam"
 * before "likelihood"; NormalLikelihood<float>, include/mcl_3dl/nd.h:41-58. */
double orc_measure_update(void* h, const float* poses, const float* odom_err_integ_lin, float* weight_inout,
                          size_t n_p, const float* scan_lik_xyz, size_t n_s, const float* scan_beam_xyz,
                          const uint32_t* scan_beam_label, size_t n_b, const float* origins_xyz, size_t n_o,
                          float odom_err_integ_lin_sigma, float* out_lik, float* out_beam, float* out_quality,
                          float* entropy, float* match_ratio_min_out, float* match_ratio_max_out, int* restored_out)
{
  const Oracle* o = (const Oracle*)h;
  (void)n_o;
  Caster c;
  memset(&c, 0, sizeof(c));
  caster_prepare(&c, o);
  if (o->use_dda && n_b)
    dda_update(&c.dda, o);
  const size_t nmax = n_s > n_b ? n_s : n_b;
  float* tmp = (float*)malloc(sizeof(float) * 3 * (nmax ? nmax : 1));
  float* ret = (float*)malloc(sizeof(float) * (n_p ? n_p : 1));
  float match_ratio_min = 1.0f, match_ratio_max = 0.0f;
  const float sigma = odom_err_integ_lin_sigma;
  const float nd_a = (float)(1.0 / sqrt(2.0 * M_PI * sigma * sigma)); /* nd.h:46 */
  const float nd_sq2 = (float)(sigma * sigma * 2.0);                   /* nd.h:47 */
  const double t0 = now_sec();
  for (size_t i = 0; i < n_p; ++i)
  {
    float likelihood = 1;
    float lb, qb, ll, ql;
    beam_measure1(&c, o, &poses[7 * i], scan_beam_xyz, scan_beam_label, n_b, origins_xyz, tmp, &lb, &qb);
    likelihood *= lb;
    likelihood_measure1(o, &poses[7 * i], scan_lik_xyz, n_s, tmp, &ll, &ql);
    likelihood *= ll;
    if (out_lik)
      out_lik[i] = ll;
    if (out_beam)
      out_beam[i] = lb;
    if (out_quality)
      out_quality[i] = ql;
    if (match_ratio_min > ql)
      match_ratio_min = ql;
    if (match_ratio_max < ql)
      match_ratio_max = ql;
    float x = 0.0f;
    if (odom_err_integ_lin)
      x = v3_norm(v3(odom_err_integ_lin[3 * i], odom_err_integ_lin[3 * i + 1], odom_err_integ_lin[3 * i + 2]));
    const float odom_error = nd_a * expf(-x * x / nd_sq2); /* nd.h:51 */
    ret[i] = likelihood * odom_error;
  }
  float e;
  const int restored = pf_measure_core(weight_inout, ret, n_p, &e);
  const double dt = now_sec() - t0;
  if (entropy)
    *entropy = e;
  if (restored_out)
    *restored_out = restored;
  if (match_ratio_min_out)
    *match_ratio_min_out = match_ratio_min;
  if (match_ratio_max_out)
    *match_ratio_max_out = match_ratio_max;
  free(tmp);
  free(ret);
  dda_free(&c.dda);
  return dt;
}

/* ------------------------------------------------------------------------------------------------
 * expectationBiased / max / covariance — pf.h:280-390 with ParticleWeightedMeanQuat (state_6dof.h:316-355)
 * ---------------------------------------------------------------------------------------------- */
static V3 v3_cross(V3 a, V3 q) /* vec3.h:145-150 */
{
  return v3(a.y * q.z - a.z * q.y, a.z * q.x - a.x * q.z, a.x * q.y - a.y * q.x);
}

/* Quat(const Vec3& forward, const Vec3& up_raw), quat.h:61-80 */
static Q4 q_from_front_up(V3 forward, V3 up_raw)
{
  const V3 xv = v3_normalized(forward);
  const V3 yv = v3_normalized(v3_cross(up_raw, xv));
  const V3 zv = v3_normalized(v3_cross(xv, yv));
  Q4 q;
  q.w = (float)(sqrt(fmax(0.0, 1.0 + xv.x + yv.y + zv.z)) / 2.0);
  q.x = (float)(sqrt(fmax(0.0, 1.0 + xv.x - yv.y - zv.z)) / 2.0);
  q.y = (float)(sqrt(fmax(0.0, 1.0 - xv.x + yv.y - zv.z)) / 2.0);
  q.z = (float)(sqrt(fmax(0.0, 1.0 - xv.x - yv.y + zv.z)) / 2.0);
  if (zv.y - yv.z > 0)
    q.x = -q.x;
  if (xv.z - zv.x > 0)
    q.y = -q.y;
  if (yv.x - xv.y > 0)
    q.z = -q.z;
  return q;
}

/* Quat::getRPY, quat.h:188-203 */
static V3 q_get_rpy(Q4 q)
{
  const float ysq = q.y * q.y;
  const float t0 = (float)(-2.0 * (ysq + q.z * q.z) + 1.0);
  const float t1 = (float)(+2.0 * (q.x * q.y + q.w * q.z));
  const float t2 = (float)fmax(-1.0, fmin(1.0, -2.0 * (q.x * q.z - q.w * q.y)));
  const float t3 = (float)(+2.0 * (q.y * q.z + q.w * q.x));
  const float t4 = (float)(-2.0 * (q.x * q.x + ysq) + 1.0);
  return v3(atan2f(t3, t4), asinf(t2), atan2f(t1, t0));
}

typedef struct
{
  float p_sum;
  V3 pos, front, up;
} WMean; /* ParticleWeightedMeanQuat */

static void wmean_add(WMean* m, const float* pose7, float prob) /* state_6dof.h:330-343 */
{
  const Q4 rot = { pose7[3], pose7[4], pose7[5], pose7[6] };
  m->p_sum += prob;
  m->pos = v3_add(m->pos, v3_scale(v3(pose7[0], pose7[1], pose7[2]), prob));
  m->front = v3_add(m->front, v3_scale(q_rot(rot, v3(1.0f, 0.0f, 0.0f)), prob));
  m->up = v3_add(m->up, v3_scale(q_rot(rot, v3(0.0f, 0.0f, 1.0f)), prob));
}

static void wmean_get(const WMean* m, float* out7) /* state_6dof.h:345-350 */
{
  const Q4 q = q_from_front_up(m->front, m->up);
  out7[0] = m->pos.x / m->p_sum;
  out7[1] = m->pos.y / m->p_sum;
  out7[2] = m->pos.z / m->p_sum;
  out7[3] = q.x;
  out7[4] = q.y;
  out7[5] = q.z;
  out7[6] = q.w;
}

void orc_expectation(const float* poses, const float* weights, const float* bias, size_t n, float* out_mean7,
                     int* out_max_index, int* out_max_biased_index)
{
  WMean m;
  memset(&m, 0, sizeof(m));
  for (size_t i = 0; i < n; ++i) /* expectationBiased, pf.h:294-303 */
    wmean_add(&m, &poses[7 * i], weights[i] * (bias ? bias[i] : 1.0f));
  wmean_get(&m, out_mean7);
  size_t im = 0, ib = 0; /* max / maxBiased, pf.h:361-390: strict <, so the first maximum wins */
  float mp = weights[0], mb = weights[0] * (bias ? bias[0] : 1.0f);
  for (size_t i = 0; i < n; ++i)
  {
    if (mp < weights[i])
    {
      mp = weights[i];
      im = i;
    }
    const float pb = weights[i] * (bias ? bias[i] : 1.0f);
    if (mb < pb)
    {
      mb = pb;
      ib = i;
    }
  }
  *out_max_index = (int)im;
  *out_max_biased_index = (int)ib;
}

/* State6DOF::covElement, state_6dof.h:162-184 */
static float cov_element(const float* s7, const float* e7, V3 rpy, V3 exp_rpy, size_t j, size_t k)
{
  float val = 1.0f, diff = 0.0f;
  const size_t idx[2] = { j, k };
  const float r[3] = { rpy.x, rpy.y, rpy.z }, er[3] = { exp_rpy.x, exp_rpy.y, exp_rpy.z };
  for (int t = 0; t < 2; ++t)
  {
    const size_t i = idx[t];
    if (i < 3)
    {
      diff = s7[i] - e7[i];
    }
    else
    {
      diff = r[i - 3] - er[i - 3];
      while (diff > M_PI)
        diff -= 2 * M_PI;
      while (diff < -M_PI)
        diff += 2 * M_PI;
    }
    val *= diff;
  }
  return val;
}

void orc_covariance(const float* poses, const float* weights, size_t n, float* out_cov36, float* out_mean7)
{
  /* expectation(1.0), pf.h:280-293: stops once the running float sum exceeds pass_ratio */
  WMean m;
  memset(&m, 0, sizeof(m));
  for (size_t i = 0; i < n; ++i)
  {
    wmean_add(&m, &poses[7 * i], weights[i]);
    if (m.p_sum > 1.0f)
      break;
  }
  float e7[7];
  wmean_get(&m, e7);
  memcpy(out_mean7, e7, sizeof(e7));
  /* covariance, pf.h:308-359 */
  float p_sum = 0;
  size_t p_num = 0;
  for (size_t i = 0; i < n; ++i)
  {
    p_num++;
    p_sum += weights[i];
    if (p_sum > 1.0f)
      break;
  }
  float cov[6][6];
  memset(cov, 0, sizeof(cov));
  const Q4 eq = { e7[3], e7[4], e7[5], e7[6] };
  const V3 exp_rpy = q_get_rpy(eq);
  p_sum = 0.0f;
  for (size_t i = 0; i < p_num; ++i)
  {
    const float* s = &poses[7 * i];
    const Q4 sq = { s[3], s[4], s[5], s[6] };
    const V3 rpy = q_get_rpy(sq);
    p_sum += weights[i];
    for (size_t j = 0; j < 6; ++j)
      for (size_t k = j; k < 6; ++k)
      {
        cov[j][k] += cov_element(s, e7, rpy, exp_rpy, j, k) * weights[i];
        cov[k][j] = cov[j][k];
      }
  }
  for (size_t j = 0; j < 6; ++j)
    for (size_t k = 0; k < 6; ++k)
      out_cov36[6 * j + k] = cov[k][j] / p_sum;
}

/* ------------------------------------------------------------------------------------------------
 * pf::ParticleFilter::resample / resizeParticle — include/mcl_3dl/pf.h:187-225, 399-436
 *
 * The reference sorts a copy of the particle array by accumulated probability with std::sort. The accumulated
 * probabilities are already ascending, but particles of weight 0 TIE with their predecessor, and libstdc++'s
 * introsort is not stable: which particle of a tie group ends up first (and is therefore the one lower_bound
 * picks) is decided by its partition swaps. To stay bit-identical, std::sort itself is restated below
 * (third-party: libstdc++ <bits/stl_algo.h>, GCC 11: __introsort_loop, threshold 16, median-of-3 moved to
 * first, __unguarded_partition, __final_insertion_sort; the heapsort fallback is never reached for
 * ascending input and is restated for completeness). Pinned against the real std::sort through oracle/_ref.
 * ---------------------------------------------------------------------------------------------- */
typedef struct
{
  float key;    /* accum_probability_ */
  uint32_t idx; /* which particle */
} SortItem;

#define LESS(a, b) ((a).key < (b).key) /* Particle::operator<, pf.h:104-107 */

static void si_swap(SortItem* a, SortItem* b)
{
  SortItem t = *a;
  *a = *b;
  *b = t;
}

static void si_move_median_to_first(SortItem* result, SortItem* a, SortItem* b, SortItem* c)
{
  if (LESS(*a, *b))
  {
    if (LESS(*b, *c))
      si_swap(result, b);
    else if (LESS(*a, *c))
      si_swap(result, c);
    else
      si_swap(result, a);
  }
  else if (LESS(*a, *c))
    si_swap(result, a);
  else if (LESS(*b, *c))
    si_swap(result, c);
  else
    si_swap(result, b);
}

static SortItem* si_unguarded_partition(SortItem* first, SortItem* last, SortItem* pivot)
{
  for (;;)
  {
    while (LESS(*first, *pivot))
      ++first;
    --last;
    while (LESS(*pivot, *last))
      --last;
    if (!(first < last))
      return first;
    si_swap(first, last);
    ++first;
  }
}

static void si_adjust_heap(SortItem* first, long hole, long len, SortItem value)
{
  const long top = hole;
  long child = hole;
  while (child < (len - 1) / 2)
  {
    child = 2 * (child + 1);
    if (LESS(first[child], first[child - 1]))
      child--;
    first[hole] = first[child];
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2)
  {
    child = 2 * (child + 1);
    first[hole] = first[child - 1];
    hole = child - 1;
  }
  long parent = (hole - 1) / 2; /* __push_heap */
  while (hole > top && LESS(first[parent], value))
  {
    first[hole] = first[parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  first[hole] = value;
}

static void si_heapsort(SortItem* first, SortItem* last) /* __partial_sort(first, last, last) */
{
  const long len = last - first;
  if (len >= 2)
    for (long parent = (len - 2) / 2;; --parent)
    {
      si_adjust_heap(first, parent, len, first[parent]);
      if (parent == 0)
        break;
    }
  while (last - first > 1)
  {
    --last;
    SortItem value = *last;
    *last = *first;
    si_adjust_heap(first, 0, last - first, value);
  }
}

static void si_introsort_loop(SortItem* first, SortItem* last, long depth_limit)
{
  while (last - first > 16)
  {
    if (depth_limit == 0)
    {
      si_heapsort(first, last);
      return;
    }
    --depth_limit;
    SortItem* mid = first + (last - first) / 2;
    si_move_median_to_first(first, first + 1, mid, last - 1);
    SortItem* cut = si_unguarded_partition(first + 1, last, first);
    si_introsort_loop(cut, last, depth_limit);
    last = cut;
  }
}

static void si_unguarded_linear_insert(SortItem* last)
{
  SortItem val = *last;
  SortItem* next = last - 1;
  while (LESS(val, *next))
  {
    *last = *next;
    last = next;
    --next;
  }
  *last = val;
}

static void si_insertion_sort(SortItem* first, SortItem* last)
{
  if (first == last)
    return;
  for (SortItem* i = first + 1; i != last; ++i)
  {
    if (LESS(*i, *first))
    {
      SortItem val = *i;
      memmove(first + 1, first, (size_t)(i - first) * sizeof(SortItem));
      *first = val;
    }
    else
      si_unguarded_linear_insert(i);
  }
}

static void si_std_sort(SortItem* first, SortItem* last)
{
  if (first == last)
    return;
  long n = last - first, lg = 0;
  while (n > 1)
  {
    n >>= 1;
    ++lg;
  }
  si_introsort_loop(first, last, lg * 2);
  if (last - first > 16)
  {
    si_insertion_sort(first, first + 16);
    for (SortItem* i = first + 16; i != last; ++i)
      si_unguarded_linear_insert(i);
  }
  else
    si_insertion_sort(first, last);
}

static size_t si_lower_bound(const SortItem* a, size_t from, size_t n, float pscan)
{
  size_t lo = from, len = n - from; /* std::lower_bound */
  while (len > 0)
  {
    const size_t half = len >> 1, mid = lo + half;
    if (a[mid].key < pscan)
    {
      lo = mid + 1;
      len = len - half - 1;
    }
    else
      len = half;
  }
  return lo;
}

float orc_resample_pstep(const float* weight, size_t n, size_t n_out)
{
  float accum = 0;
  for (size_t i = 0; i < n; ++i)
    accum += weight[i]; /* pf.h:193-197 / 401-405 */
  return accum / n_out; /* :202 (n_out == n) / :410 */
}

/* mode 0: resample (pf.h:191-224), mode 1: resizeParticle (pf.h:399-436).
 * out_source[i] = index of the particle whose state slot i receives; out_dup[i] = 1 when the reference adds noise. */
void orc_resample_plan(const float* weight, size_t n, size_t n_out, int mode, float initial_p, uint32_t* out_source,
                       uint8_t* out_dup)
{
  SortItem* dup = (SortItem*)malloc(sizeof(SortItem) * (n ? n : 1));
  float accum = 0;
  for (size_t i = 0; i < n; ++i)
  {
    accum += weight[i];
    dup[i].key = accum;
    dup[i].idx = (uint32_t)i;
  }
  si_std_sort(dup, dup + n); /* std::sort(particles_dup_), :200 / :408 */
  const float pstep = accum / n_out;
  float pscan = 0;
  size_t it = 0, it_prev = 0;
  for (size_t i = 0; i < n_out; ++i)
  {
    if (mode == 0)
      pscan = pstep * i + initial_p; /* :209 */
    else
      pscan += pstep; /* :421 */
    it = si_lower_bound(dup, it, n, pscan);
    out_dup[i] = 0;
    if (it == n)
    {
      out_source[i] = dup[it_prev].idx; /* :212-216 / :425-429 */
      continue;
    }
    if (mode == 0 && it == it_prev)
      out_dup[i] = 1; /* :217-221 */
    out_source[i] = dup[it].idx;
    it_prev = it;
  }
  free(dup);
}

/* State6DOF::operator+ (state_6dof.h:248-260) then normalize() (:150-153) for duplicated particles; plain copies otherwise. */
void orc_resample_apply(const float* state13_in, const uint32_t* source, const uint8_t* dupf, const float* noise13,
                        size_t n_out, float* state13_out)
{
  size_t d = 0;
  for (size_t i = 0; i < n_out; ++i)
  {
    const float* s = &state13_in[13 * (size_t)source[i]];
    float* o = &state13_out[13 * i];
    if (!dupf[i])
    {
      memcpy(o, s, sizeof(float) * 13);
      continue;
    }
    const float* a = &noise13[13 * d++];
    for (int k = 0; k < 13; ++k)
      if (k < 3 || k > 6)
        o[k] = s[k] + a[k];
    const Q4 ar = { a[3], a[4], a[5], a[6] }, sr = { s[3], s[4], s[5], s[6] };
    const Q4 r = q_normalized(q_mul(ar, sr)); /* ret.rot_ = a.rot_ * rot_; then rot_.normalize() */
    o[3] = r.x;
    o[4] = r.y;
    o[5] = r.z;
    o[6] = r.w;
  }
}

/* ------------------------------------------------------------------------------------------------
 * Workload statistics for the algorithmic-bytes accounting — oracle-only helpers.
 * ---------------------------------------------------------------------------------------------- */
void orc_count_neighbourhood(void* h, const float* poses, size_t n_p, const float* scan_xyz, size_t n_s,
                             double* sum_k, double* sum_found)
{
  const Oracle* o = (const Oracle*)h;
  float* tmp = (float*)malloc(sizeof(float) * 3 * (n_s ? n_s : 1));
  const double r = o->match_dist_min;
  double K = 0, F = 0;
  for (size_t ip = 0; ip < n_p; ++ip)
  {
    transform_points(&poses[7 * ip], scan_xyz, n_s, tmp);
    for (size_t i = 0; i < n_s; ++i)
    {
      float q[3];
      vectorize(o, &tmp[3 * i], q);
      long cq[3];
      int lo[3], hi[3], skip = 0;
      for (int a = 0; a < 3; ++a)
      {
        cq[a] = (long)floor(((double)q[a] - o->g_origin[a]) / r);
        const double l = floor(((double)q[a] - 2 * r - o->g_origin[a]) / o->g_cell) - 1;
        const double u = floor(((double)q[a] + 2 * r - o->g_origin[a]) / o->g_cell) + 1;
        if (u < 0 || l > o->g_dim[a] - 1)
          skip = 1;
        lo[a] = (int)(l < 0 ? 0 : l);
        hi[a] = (int)(u > o->g_dim[a] - 1 ? o->g_dim[a] - 1 : u);
      }
      if (!skip)
        for (int z = lo[2]; z <= hi[2]; ++z)
          for (int y = lo[1]; y <= hi[1]; ++y)
          {
            const size_t row = ((size_t)z * o->g_dim[1] + y) * o->g_dim[0];
            for (uint32_t k = o->g_start[row + lo[0]]; k < o->g_start[row + hi[0] + 1]; ++k)
            {
              const float* p = &o->sxyz[3 * o->g_ids[k]];
              int in = 1;
              for (int a = 0; a < 3; ++a)
              {
                const long cp = (long)floor(((double)p[a] - o->g_origin[a]) / r);
                if (cp < cq[a] - 1 || cp > cq[a] + 1)
                  in = 0;
              }
              K += in;
            }
          }
      int id;
      float sq;
      F += radius_search1(o, &tmp[3 * i], o->match_dist_min, &id, &sq);
    }
  }
  free(tmp);
  *sum_k = K;
  *sum_found = F;
}

void orc_count_dda(void* h, const float* poses, size_t n_p, const float* scan_xyz, const uint32_t* scan_label,
                   size_t n_b, const float* origins_xyz, size_t n_o, double* steps, double* occupied, double* tested)
{
  const Oracle* o = (const Oracle*)h;
  (void)n_o;
  Caster c;
  memset(&c, 0, sizeof(c));
  caster_prepare(&c, o);
  float* tmp = (float*)malloc(sizeof(float) * 3 * (n_b ? n_b : 1));
  for (size_t i = 0; i < n_p; ++i)
  {
    float l, q;
    beam_measure1(&c, o, &poses[7 * i], scan_xyz, scan_label, n_b, origins_xyz, tmp, &l, &q);
  }
  *steps = c.dda.n_steps;
  *occupied = c.dda.n_occupied;
  *tested = c.dda.n_tested;
  free(tmp);
  dda_free(&c.dda);
}

/* ------------------------------------------------------------------------------------------------
 * lifecycle / parameters
 * ---------------------------------------------------------------------------------------------- */
void* orc_create(float chunk_length, float max_search_radius)
{
  Oracle* o = (Oracle*)calloc(1, sizeof(Oracle));
  o->pos_to_chunk = (float)(1.0 / chunk_length); /* chunked_kdtree.h:95 */
  o->chunk_length = chunk_length;
  o->max_search_radius = max_search_radius;
  /* parameter defaults, include/mcl_3dl/parameters.h:67-112 */
  o->match_dist_min = 0.2f;
  o->match_dist_flat = 0.05f;
  o->match_weight = 5.0f;
  o->map_grid[0] = o->map_grid[1] = o->map_grid[2] = 0.1f;
  o->dda_grid_size = 0.2f;
  o->ray_angle_half = (float)(0.25 * M_PI / 180.0);
  o->hit_range = 0.3f;
  o->beam_likelihood_min = 0.2f;
  o->ang_total_ref = (float)(M_PI / 6.0);
  o->beam_num_points = 3;
  o->filter_label_max = 0xFFFFFFFFu;
  o->short_only = 1;
  o->use_dda = 0;
  beam_refresh(o);
  return o;
}

void orc_destroy(void* h)
{
  Oracle* o = (Oracle*)h;
  if (!o)
    return;
  free_map(o);
  free(o);
}

int orc_max_threads(void)
{
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

void orc_set_likelihood_params(void* h, float match_dist_min, float match_dist_flat, float match_weight,
                               uint32_t num_points, uint32_t num_points_global, float clip_near, float clip_far,
                               float clip_z_min, float clip_z_max)
{
  Oracle* o = (Oracle*)h;
  o->match_dist_min = match_dist_min;
  o->match_dist_flat = match_dist_flat;
  o->match_weight = match_weight;
  (void)num_points;
  (void)num_points_global;
  (void)clip_near;
  (void)clip_far;
  (void)clip_z_min;
  (void)clip_z_max; /* filter() inputs; not on the measure() path */
}

void orc_set_beam_params(void* h, float map_grid_x, float map_grid_y, float map_grid_z, float dda_grid_size,
                         float ray_angle_half, float hit_range, float beam_likelihood_min, uint32_t num_points,
                         uint32_t num_points_global, float ang_total_ref, uint32_t filter_label_max,
                         int add_penalty_short_only_mode, int use_raycast_using_dda, float clip_near, float clip_far,
                         float clip_z_min, float clip_z_max)
{
  Oracle* o = (Oracle*)h;
  o->map_grid[0] = map_grid_x;
  o->map_grid[1] = map_grid_y;
  o->map_grid[2] = map_grid_z;
  o->dda_grid_size = dda_grid_size;
  o->ray_angle_half = ray_angle_half;
  o->hit_range = hit_range;
  o->beam_likelihood_min = beam_likelihood_min;
  o->beam_num_points = num_points;
  o->ang_total_ref = ang_total_ref;
  o->filter_label_max = filter_label_max;
  o->short_only = add_penalty_short_only_mode != 0;
  o->use_dda = use_raycast_using_dda != 0;
  (void)num_points_global;
  (void)clip_near;
  (void)clip_far;
  (void)clip_z_min;
  (void)clip_z_max;
  beam_refresh(o);
}
