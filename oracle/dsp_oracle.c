/*
 * dsp_oracle.c -- CPU ORACLE (test infrastructure, NOT the product).
 * See dsp_oracle.h for scope, the "parity unpinned" statement and who may
 * call this.  References are to /root/reference/include/dsp_dynamic.h unless
 * another file is named.
 *
 * Build for checking:   gcc -O2 -ffp-contract=off -fno-fast-math   (strict IEEE)
 * Build for CPU timing: the reference's own flags, CMakeLists.txt:4
 *                       (-O3 -ftree-vectorize -ffast-math -march=native)
 */
#define _GNU_SOURCE
#include "dsp_oracle.h"

#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define PSTRIDE 9 /* record {flag,vx,vy,vz,px,py,pz,weight,update_time} :114-116 */

typedef struct cluster_feature { /* ClusterFeature :98-109 */
    float cx, cy, cz;
    int point_num;
    int match_seq;
    float vx, vy, vz, v, intensity;
} cluster_feature;

struct dsp_oracle {
    dspo_config cfg;
    /* derived sizes :58-66 */
    int np_h, np_v, np;     /* observation_pyramid_num_{h,v}, observation_pyramid_num */
    int V;                  /* VOXEL_NUM */
    int slots;              /* SAFE_PARTICLE_NUM_VOXEL */
    int capp;               /* SAFE_PARTICLE_NUM_PYRAMID */
    int rdim;               /* voxels_objects_number_dimension :119 */
    float half_x, half_y, half_z; /* map_length_*_half :528-530 */
    float res;
    /* parameters :145-168 */
    float p_stddev, v_stddev, sigma_ob, kappa, P_detection;
    float update_time;
    int update_counter;
    float expected_new_born_objects;
    float new_born_particle_weight;
    int new_born_particle_number_each_point;
    float new_born_each_object_weight;
    float voxel_filtered_resolution; /* :132 */
    /* state */
    float* particles;  /* [V][slots][9] */
    float* results;    /* [V][rdim]     */
    int* pyr_lists;    /* [np][capp][3] */
    int* neighbors;    /* [np][nbs]     */
    int nn, nbs;       /* neighbourhood radius, table stride */
    float occl_margin;
    float* obs;        /* [np][100][5]  */
    int* obs_count;    /* [np] */
    float* obs_maxlen; /* [np] */
    float* bp_ori_h;   /* [np_h+1][3] pyramid_BPnorm_params_ori_h :506 */
    float* bp_ori_v;   /* [np_v+1][3] */
    float* bp_h;       /* rotated :510-511 */
    float* bp_v;
    float quat[4];     /* sensor_rotation_quaternion :503 */
    float pdf[DSPO_PDF_LUT_SIZE];
    /* randomness */
    const float* p_tab;
    const float* v_tab;
    int tab_n;
    int p_cur, v_cur;
    const int* r_tab;
    int r_n, r_cur;
    /* function statics of update() :187-190 */
    int have_last;
    float last_px, last_py, last_pz;
    double last_stamp;
    float current_position[3];  /* :131 */
    float delt_t_from_last;     /* :133 */
    /* function statics of mapAddNewBornParticlesByObservation :808-811 */
    int nb_statics_init;
    int min_static_nb, model_generated_nb;
    /* clouds */
    float* cloud_view; int cloud_view_n, cloud_view_cap;      /* cloud_in_current_view_rotated :130 */
    dspo_vpoint* birth; int birth_n, birth_cap;               /* input_cloud_with_velocity :134 */
    int use_vel_est;
    const int* nstatic_override;
    cluster_feature* last_clusters; int last_clusters_n;      /* clusters_feature_vector_dynamic_last :1401 */
};

/* ------------------------------------------------------------------ config */
void dspo_default_config(dspo_config* c) { /* :38-50 */
    memset(c, 0, sizeof(*c));
    c->nx = 66; c->ny = 66; c->nz = 40;
    c->voxel_resolution = 0.15f;
    c->angle_resolution = 3;
    c->max_particle_num_voxel = 9;
    c->half_fov_h = 42; c->half_fov_v = 24;
    c->prediction_times = 6;
    const float t[6] = {0.05f, 0.2f, 0.5f, 1.f, 1.5f, 2.f};
    memcpy(c->prediction_future_time, t, sizeof(t));
}

/* ---------------------------------------------------------- small helpers */
#define PART(o, v, s) ((o)->particles + ((size_t)(v) * (o)->slots + (s)) * PSTRIDE)
#define RES(o, v) ((o)->results + (size_t)(v) * (o)->rdim)
#define PYR(o, b, j) ((o)->pyr_lists + ((size_t)(b) * (o)->capp + (j)) * 3)
#define OBS(o, b, j) ((o)->obs + ((size_t)(b) * DSPO_OBS_MAX_PER_PYRAMID + (j)) * 5)

/* standardNormalPDF :1282-1286.  sqrtf(2*pi/2) = sqrt(pi): the reference's
 * constant is 1/sqrt(pi), not 1/sqrt(2 pi). */
static float standard_normal_pdf(float value) {
    const float pi_2 = 1.57079632679489661923f; /* M_PI_2f32 */
    return (1.f / (sqrtf(2.f * pi_2))) * expf(-powf(value, 2) / (2));
}

/* queryNormalPDF :1294-1301 */
float dspo_query_normal_pdf(const dsp_oracle* o, float x, float mu, float sigma) {
    float corrected_x = (x - mu) / sigma;
    if (corrected_x > 9.9f) corrected_x = 9.9f;
    else if (corrected_x < -9.9f) corrected_x = -9.9f;
    return o->pdf[(int)(corrected_x * 1000 + 10000)];
}
const float* dspo_pdf_lut(const dsp_oracle* o) { return o->pdf; }

/* rotateVectorByQuaternion :1303-1322: att * (0,v) * att.inverse() with Eigen.
 * Eigen is not in /root/reference (version unpinned by the reference's
 * readme.md:21-24); the Hamilton product and inverse = conjugate/squaredNorm
 * are restated in Eigen's generic (non-vectorised) operand order. */
static void quat_mul(const float a[4], const float b[4], float r[4]) { /* w,x,y,z */
    r[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    r[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    r[2] = a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3];
    r[3] = a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1];
}
void dspo_rotate_vector(const float v[3], const float q[4], float out[3]) {
    float vq[4] = {0.f, v[0], v[1], v[2]};
    float n2 = q[1] * q[1] + q[2] * q[2] + q[3] * q[3] + q[0] * q[0];
    float inv[4] = {q[0] / n2, -q[1] / n2, -q[2] / n2, -q[3] / n2};
    float t[4], r[4];
    quat_mul(q, vq, t);
    quat_mul(t, inv, r);
    out[0] = r[1]; out[1] = r[2]; out[2] = r[3];
}

/* vectorMultiply :1324-1326 */
static float vdot(float x1, float y1, float z1, float x2, float y2, float z2) {
    return x1 * x2 + y1 * y2 + z1 * z2;
}

/* ifInPyramidsArea :1329-1339 */
int dspo_in_pyramids_area(const dsp_oracle* o, float x, float y, float z) {
    const float* h0 = o->bp_h;
    const float* hN = o->bp_h + 3 * o->np_h;
    const float* v0 = o->bp_v;
    const float* vN = o->bp_v + 3 * o->np_v;
    if (vdot(x, y, z, h0[0], h0[1], h0[2]) >= 0.f && vdot(x, y, z, hN[0], hN[1], hN[2]) <= 0.f &&
        vdot(x, y, z, v0[0], v0[1], v0[2]) <= 0.f && vdot(x, y, z, vN[0], vN[1], vN[2]) >= 0.f)
        return 1;
    return 0;
}
/* findPointPyramidHorizontalIndex :1341-1353 */
int dspo_pyramid_h(const dsp_oracle* o, float x, float y, float z) {
    float last = 1.f;
    for (int i = 0; i < o->np_h; i++) {
        const float* n = o->bp_h + 3 * (i + 1);
        float t = vdot(x, y, z, n[0], n[1], n[2]);
        if (last * t <= 0.f) return i;
        last = t;
    }
    return -1;
}
/* findPointPyramidVerticalIndex :1355-1367 */
int dspo_pyramid_v(const dsp_oracle* o, float x, float y, float z) {
    float last = -1.f;
    for (int j = 0; j < o->np_v; j++) {
        const float* n = o->bp_v + 3 * (j + 1);
        float t = vdot(x, y, z, n[0], n[1], n[2]);
        if (last * t <= 0.f) return j;
        last = t;
    }
    return -1;
}

/* ifParticleIsOut :1118-1125 */
static int particle_is_out(const dsp_oracle* o, float px, float py, float pz) {
    if (px >= o->half_x || px <= -o->half_x || py >= o->half_y || py <= -o->half_y ||
        pz >= o->half_z || pz <= -o->half_z)
        return 1;
    return 0;
}
/* getParticleVoxelsIndex :1076-1088 (and the Particle overload :1062-1074) */
int dspo_voxel_index(const dsp_oracle* o, float px, float py, float pz, int* index) {
    if (particle_is_out(o, px, py, pz)) return 0;
    int x = (int)((px + o->half_x) / o->res);
    int y = (int)((py + o->half_y) / o->res);
    int z = (int)((pz + o->half_z) / o->res);
    *index = z * o->cfg.ny * o->cfg.nx + y * o->cfg.nx + x;
    if (*index < 0 || *index >= o->V) return 0;
    return 1;
}
/* getVoxelPositionFromIndex :1090-1107 */
void dspo_voxel_center(const dsp_oracle* o, int index, float* px, float* py, float* pz) {
    int zc = o->cfg.ny * o->cfg.nx, yc = o->cfg.nx;
    int zi = index / zc;
    int rest = index - zi * zc;
    int yi = rest / yc;
    int xi = rest - yi * yc;
    float cx = -o->half_x + o->res * 0.5f;
    float cy = -o->half_y + o->res * 0.5f;
    float cz = -o->half_z + o->res * 0.5f;
    *px = (float)xi * o->res + cx;
    *py = (float)yi * o->res + cy;
    *pz = (float)zi * o->res + cz;
}

/* findPyramidNeighborIndexInFOV :1128-1147 */
static void find_neighbors(const dsp_oracle* o, int index_ori, int* num, int* out) {
    int h0 = index_ori / o->np_v, v0 = index_ori % o->np_v;
    *num = 0;
    const int N = o->nn;   /* 1 in dsp_dynamic.h; PYRAMID_NEIGHBOR_N in dsp_dynamic_multiple_neighbors.h:1135-1136 */
    for (int i = -N; i <= N; ++i)
        for (int j = -N; j <= N; ++j) {
            int h = h0 + i, v = v0 + j;
            if (h >= 0 && h < o->np_h && v >= 0 && v < o->np_v) {
                out[*num] = h * o->np_v + v;
                ++*num;
            }
        }
}
const int* dspo_neighbor_table(const dsp_oracle* o) { return o->neighbors; }

/* getPositionGaussianZeroCenter :1162-1169 / getVelocityGaussianZeroCenter :1171-1178 */
static float draw_p(dsp_oracle* o) {
    float d = o->p_tab ? o->p_tab[o->p_cur] : 0.f;
    o->p_cur += 1;
    if (o->p_cur >= o->tab_n) o->p_cur = 0;
    return d;
}
static float draw_v(dsp_oracle* o) {
    float d = o->v_tab ? o->v_tab[o->v_cur] : 0.f;
    o->v_cur += 1;
    if (o->v_cur >= o->tab_n) o->v_cur = 0;
    return d;
}
static int next_rand(dsp_oracle* o) {
    if (!o->r_tab) return rand();
    int r = o->r_tab[o->r_cur];
    o->r_cur += 1;
    if (o->r_cur >= o->r_n) o->r_cur = 0;
    return r;
}
/* generateRandomFloat :1551-1553 */
float dspo_generate_random_float(dsp_oracle* o, float lo, float hi) {
    return lo + (float)(next_rand(o)) / ((float)(RAND_MAX / (hi - lo)));
}

/* --------------------------------------------------------------- lifecycle */
dsp_oracle* dspo_create(const dspo_config* cfg) {
    dsp_oracle* o = (dsp_oracle*)calloc(1, sizeof(dsp_oracle));
    if (!o) return NULL;
    o->cfg = *cfg;
    const int A = cfg->angle_resolution;
    o->np_h = cfg->half_fov_h * 2 / A;  /* :58 */
    o->np_v = cfg->half_fov_v * 2 / A;  /* :59 */
    o->np = o->np_h * o->np_v;          /* :60 */
    o->V = cfg->nx * cfg->ny * cfg->nz; /* :62 */
    int pyramid_num = 360 * 180 / A / A;                                         /* :63 */
    int safe_particle_num = (int)((double)o->V * cfg->max_particle_num_voxel + 1e5); /* :64 */
    o->nn = cfg->pyramid_neighbor_n > 0 ? cfg->pyramid_neighbor_n : 1;          /* 3x3; dsp_dynamic_multiple_neighbors.h:43 uses 2 */
    o->nbs = (2 * o->nn + 1) * (2 * o->nn + 1) + 1;                              /* neighbour table stride, :126-127 */
    o->occl_margin = 0.3f;                                                       /* obstacle_thickness_for_occlusion :70 */
    o->slots = cfg->max_particle_num_voxel * (cfg->safe_particle_factor > 0 ? cfg->safe_particle_factor : 2); /* :65; x5 in dsp_static.h:63 */
    o->capp = safe_particle_num / pyramid_num * 2;                               /* :66 */
    o->rdim = 4 + cfg->prediction_times;                                         /* :119 */
    o->res = cfg->voxel_resolution;
    /* ctor defaults :152-168 */
    o->p_stddev = 0.2f; o->v_stddev = 0.1f; o->sigma_ob = 0.2f;
    o->kappa = 0.01f; o->P_detection = 0.95f;
    o->new_born_particle_weight = 0.04f;
    o->new_born_particle_number_each_point = 20;
    o->voxel_filtered_resolution = 0.15f;
    o->tab_n = 1; /* cursors wrap harmlessly until tables are set */
    /* setInitParameters :525-591 */
    o->half_x = (o->res * (float)cfg->nx) * 0.5f;
    o->half_y = (o->res * (float)cfg->ny) * 0.5f;
    o->half_z = (o->res * (float)cfg->nz) * 0.5f;
    o->particles = (float*)calloc((size_t)o->V * o->slots * PSTRIDE, sizeof(float));
    o->results = (float*)calloc((size_t)o->V * o->rdim, sizeof(float));
    o->pyr_lists = (int*)calloc((size_t)o->np * o->capp * 3, sizeof(int));
    o->neighbors = (int*)calloc((size_t)o->np * o->nbs, sizeof(int));
    o->obs = (float*)calloc((size_t)o->np * DSPO_OBS_MAX_PER_PYRAMID * 5, sizeof(float));
    o->obs_count = (int*)calloc((size_t)o->np, sizeof(int));
    o->obs_maxlen = (float*)calloc((size_t)o->np, sizeof(float));
    o->bp_ori_h = (float*)calloc((size_t)(o->np_h + 1) * 3, sizeof(float));
    o->bp_ori_v = (float*)calloc((size_t)(o->np_v + 1) * 3, sizeof(float));
    o->bp_h = (float*)calloc((size_t)(o->np_h + 1) * 3, sizeof(float));
    o->bp_v = (float*)calloc((size_t)(o->np_v + 1) * 3, sizeof(float));
    if (!o->particles || !o->results || !o->pyr_lists || !o->neighbors || !o->obs) {
        dspo_destroy(o);
        return NULL;
    }
    /* boundary plane normals :563-578 (float pi, float sin/cos as in C++ overloads) */
    const float pi_f = 3.14159265358979323846f;
    float ang_rad = (float)A / 180.f * pi_f; /* :543 */
    int h_end = cfg->half_fov_h / A, h_start = -h_end;
    for (int i = h_start; i <= h_end; i++) {
        o->bp_ori_h[(i + h_end) * 3 + 0] = -sinf((float)i * ang_rad);
        o->bp_ori_h[(i + h_end) * 3 + 1] = cosf((float)i * ang_rad);
        o->bp_ori_h[(i + h_end) * 3 + 2] = 0.f;
    }
    int v_end = cfg->half_fov_v / A, v_start = -v_end;
    for (int i = v_start; i <= v_end; i++) {
        o->bp_ori_v[(i + v_end) * 3 + 0] = sinf((float)i * ang_rad);
        o->bp_ori_v[(i + v_end) * 3 + 1] = 0.f;
        o->bp_ori_v[(i + v_end) * 3 + 2] = cosf((float)i * ang_rad);
    }
    memcpy(o->bp_h, o->bp_ori_h, sizeof(float) * (o->np_h + 1) * 3);
    memcpy(o->bp_v, o->bp_ori_v, sizeof(float) * (o->np_v + 1) * 3);
    o->quat[0] = 1.f;
    for (int i = 0; i < o->np; i++) /* :581-583 */
        find_neighbors(o, i, &o->neighbors[i * o->nbs], &o->neighbors[i * o->nbs + 1]);
    for (int i = 0; i < DSPO_PDF_LUT_SIZE; ++i) /* calculateNormalPDFBuffer :1288-1292 */
        o->pdf[i] = standard_normal_pdf((float)(i - 10000) * 0.001f);
    o->use_vel_est = 1;
    return o;
}

void dspo_destroy(dsp_oracle* o) {
    if (!o) return;
    free(o->particles); free(o->results); free(o->pyr_lists); free(o->neighbors);
    free(o->obs); free(o->obs_count); free(o->obs_maxlen);
    free(o->bp_ori_h); free(o->bp_ori_v); free(o->bp_h); free(o->bp_v);
    free(o->cloud_view); free(o->birth); free(o->last_clusters);
    free(o);
}

int dspo_voxel_num(const dsp_oracle* o) { return o->V; }
int dspo_slots_per_voxel(const dsp_oracle* o) { return o->slots; }
int dspo_pyramid_num(const dsp_oracle* o) { return o->np; }
int dspo_pyramid_capacity(const dsp_oracle* o) { return o->capp; }
int dspo_result_dim(const dsp_oracle* o) { return o->rdim; }

void dspo_set_prediction_variance(dsp_oracle* o, float p, float v) { o->p_stddev = p; o->v_stddev = v; } /* :355-360 */
void dspo_set_observation_stddev(dsp_oracle* o, float s) { o->sigma_ob = s; }                            /* :362 */
void dspo_set_newborn_weight(dsp_oracle* o, float w) { o->new_born_particle_weight = w; }                /* :366 */
void dspo_set_newborn_number(dsp_oracle* o, int n) { o->new_born_particle_number_each_point = n; }       /* :370 */
void dspo_set_voxel_filter_resolution(dsp_oracle* o, float r) { o->voxel_filtered_resolution = r; }      /* :380 */

void dspo_set_gaussian_tables(dsp_oracle* o, const float* p, const float* v, int n) {
    o->p_tab = p; o->v_tab = v; o->tab_n = n > 0 ? n : 1;
    o->p_cur = 0; o->v_cur = 0;
}
void dspo_set_rand_table(dsp_oracle* o, const int* r, int n) { o->r_tab = r; o->r_n = n; o->r_cur = 0; }
void dspo_set_cursors(dsp_oracle* o, int p, int v, int r) { o->p_cur = p; o->v_cur = v; o->r_cur = r; }
void dspo_get_cursors(const dsp_oracle* o, int* p, int* v, int* r) {
    if (p) *p = o->p_cur;
    if (v) *v = o->v_cur;
    if (r) *r = o->r_cur;
}

float* dspo_particles(dsp_oracle* o) { return o->particles; }
float* dspo_results(dsp_oracle* o) { return o->results; }
int* dspo_pyramid_lists(dsp_oracle* o) { return o->pyr_lists; }
float* dspo_obs(dsp_oracle* o) { return o->obs; }
int* dspo_obs_count(dsp_oracle* o) { return o->obs_count; }
float* dspo_obs_max_length(dsp_oracle* o) { return o->obs_maxlen; }
float dspo_expected_newborn(const dsp_oracle* o) { return o->expected_new_born_objects; }
void dspo_set_expected_newborn(dsp_oracle* o, float v) { o->expected_new_born_objects = v; }
void dspo_set_occlusion_margin(dsp_oracle* o, float v) { o->occl_margin = v; }
float dspo_update_time(const dsp_oracle* o) { return o->update_time; }
void dspo_use_velocity_estimator(dsp_oracle* o, int on) { o->use_vel_est = on; }
void dspo_set_current_position(dsp_oracle* o, float x, float y, float z) {
    o->current_position[0] = x; o->current_position[1] = y; o->current_position[2] = z;
}

int dspo_count_live(const dsp_oracle* o) {
    int n = 0;
    for (int v = 0; v < o->V; v++)
        for (int s = 0; s < o->slots; s++)
            if (PART(o, v, s)[0] > 0.1f) ++n;
    return n;
}

static void birth_reserve(dsp_oracle* o, int n) {
    if (n > o->birth_cap) {
        o->birth_cap = n * 2 + 64;
        o->birth = (dspo_vpoint*)realloc(o->birth, sizeof(dspo_vpoint) * o->birth_cap);
    }
}
void dspo_set_birth_cloud(dsp_oracle* o, const dspo_vpoint* pts, int n) {
    birth_reserve(o, n);
    if (n > 0) memcpy(o->birth, pts, sizeof(dspo_vpoint) * n);
    o->birth_n = n;
}
int dspo_get_birth_cloud(const dsp_oracle* o, dspo_vpoint* out, int cap) {
    int n = o->birth_n < cap ? o->birth_n : cap;
    if (out && n > 0) memcpy(out, o->birth, sizeof(dspo_vpoint) * n);
    return o->birth_n;
}

/* addAParticle :1183-1201 */
static int add_a_particle(dsp_oracle* o, float px, float py, float pz, float vx, float vy, float vz,
                          float w, int voxel_index) {
    for (int i = 0; i < o->slots; i++) {
        float* r = PART(o, voxel_index, i);
        if (r[0] < 0.1f) {
            r[0] = 15.f;
            r[1] = vx; r[2] = vy; r[3] = vz;
            r[4] = px; r[5] = py; r[6] = pz;
            r[7] = w;
            r[8] = o->update_time;
            return 1;
        }
    }
    return 0;
}

/* test helper (no counterpart in the reference): state injection -- every particle goes to the first free slot of its voxel
 * (the addAParticle rule, :1184-1185) with the given flag; particles outside the map or into a full voxel are skipped.
 * flags may be NULL (then `flag` for all).  Returns the number placed. */
int dspo_inject(dsp_oracle* o, int n, const float* px, const float* py, const float* pz, const float* vx, const float* vy,
                const float* vz, const float* w, const float* flags, float flag) {
    int placed = 0;
    for (int k = 0; k < n; k++) {
        int idx;
        if (!dspo_voxel_index(o, px[k], py[k], pz[k], &idx)) continue;
        for (int i = 0; i < o->slots; i++) {
            float* r = PART(o, idx, i);
            if (r[0] < 0.1f) {
                r[0] = flags ? flags[k] : flag;
                r[1] = vx[k]; r[2] = vy[k]; r[3] = vz[k];
                r[4] = px[k]; r[5] = py[k]; r[6] = pz[k];
                r[7] = w[k];
                ++placed;
                break;
            }
        }
    }
    return placed;
}

/* addRandomParticles :594-624 */
void dspo_add_random_particles(dsp_oracle* o, int n, float w) {
    for (int i = 0; i < n; i++) {
        float px = dspo_generate_random_float(o, -o->half_x, o->half_x);
        float py = dspo_generate_random_float(o, -o->half_y, o->half_y);
        float pz = dspo_generate_random_float(o, -o->half_z, o->half_z);
        float vx = dspo_generate_random_float(o, -1.f, 1.f);
        float vy = dspo_generate_random_float(o, -1.f, 1.f);
        float vz = dspo_generate_random_float(o, -1.f, 1.f);
        int idx;
        if (dspo_voxel_index(o, px, py, pz, &idx)) add_a_particle(o, px, py, pz, vx, vy, vz, w, idx);
    }
}

/* ------------------------------------------------------------- obs binning */
/* update() :220-293 */
int dspo_bin_points(dsp_oracle* o, int n_pts, int stride, const float* pts,
                    float qw, float qx, float qy, float qz) {
    o->quat[0] = qw; o->quat[1] = qx; o->quat[2] = qy; o->quat[3] = qz; /* :221-224 */
    for (int i = 0; i < o->np_h + 1; i++) dspo_rotate_vector(o->bp_ori_h + 3 * i, o->quat, o->bp_h + 3 * i); /* :226-228 */
    for (int j = 0; j < o->np_v + 1; j++) dspo_rotate_vector(o->bp_ori_v + 3 * j, o->quat, o->bp_v + 3 * j); /* :230-232 */
    for (int i = 0; i < o->np; i++) { /* :235-238 */
        o->obs_count[i] = 0;
        o->obs_maxlen[i] = -1.f;
    }
    o->cloud_view_n = 0; /* :240 */
    if (n_pts > o->cloud_view_cap) {
        o->cloud_view_cap = n_pts + 64;
        o->cloud_view = (float*)realloc(o->cloud_view, sizeof(float) * 3 * o->cloud_view_cap);
    }
    int iter = 0, valid_points = 0;
    for (int p = 0; p < n_pts; ++p) { /* :244-290 */
        float r[3];
        dspo_rotate_vector(pts + iter, o->quat, r);
        if (dspo_in_pyramids_area(o, r[0], r[1], r[2])) {
            float* cv = o->cloud_view + 3 * o->cloud_view_n++;
            cv[0] = r[0]; cv[1] = r[1]; cv[2] = r[2];
            int h = dspo_pyramid_h(o, r[0], r[1], r[2]);
            int v = dspo_pyramid_v(o, r[0], r[1], r[2]);
            int b = h * o->np_v + v;
            int seq = o->obs_count[b];
            float len = sqrtf(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
            float* ob = OBS(o, b, seq);
            ob[0] = r[0]; ob[1] = r[1]; ob[2] = r[2]; ob[3] = 0.f; ob[4] = len;
            if (o->obs_maxlen[b] < len) o->obs_maxlen[b] = len;
            o->obs_count[b] += 1;
            if (o->obs_count[b] >= DSPO_OBS_MAX_PER_PYRAMID) o->obs_count[b] = DSPO_OBS_MAX_PER_PYRAMID - 1; /* :282-284 */
            ++valid_points;
        }
        iter += stride;
    }
    o->expected_new_born_objects =
        o->new_born_particle_weight * (float)valid_points * (float)o->new_born_particle_number_each_point; /* :292 */
    o->new_born_each_object_weight = o->new_born_particle_weight * (float)o->new_born_particle_number_each_point; /* :293 */
    return valid_points;
}

/* -------------------------------------------------------------- prediction */
/* moveParticle :1206-1274.  returns 1, -1 (voxel full), -2 (pyramid full) */
static int move_particle(dsp_oracle* o, int new_voxel, int cur_voxel, int cur_slot, float* rec) {
    int new_slot = cur_slot;
    if (new_voxel != cur_voxel) {
        rec[0] = 0.f; /* :1210 */
        int moved = 0;
        for (int i = 0; i < o->slots; ++i) {
            float* d = PART(o, new_voxel, i);
            if (d[0] < 0.1f) {
                new_slot = i;
                moved = 1;
                d[0] = 7.f; /* :1219 */
                for (int k = 1; k < 9; ++k) d[k] = rec[k];
                break;
            }
        }
        if (!moved) return -1;
    }
    float* r = PART(o, new_voxel, new_slot);
    if (dspo_in_pyramids_area(o, r[4], r[5], r[6])) { /* :1233 */
        int h = dspo_pyramid_h(o, r[4], r[5], r[6]);
        int v = dspo_pyramid_v(o, r[4], r[5], r[6]);
        int b = h * o->np_v + v;
        int ok = 0;
        for (int j = 0; j < o->capp; j++) {
            int* e = PYR(o, b, j);
            if (e[0] == 0) {
                e[0] |= 1; e[1] = new_voxel; e[2] = new_slot;
                ok = 1;
                break;
            }
        }
        if (!ok) {
            r[0] = 0.f; /* :1257 */
            return -2;
        }
        if (fabs(r[1] * r[2] * r[3]) < 1e-6) { /* :1262 (double compare, as written) */
        } else {
            r[1] += draw_v(o);
            r[2] += draw_v(o);
            r[3] = 0.f;
        }
    }
    return 1;
}

/* mapPrediction :627-701 (LIMIT_MOVEMENT_IN_XY_PLANE == 1, :44,661-663) */
void dspo_map_prediction(dsp_oracle* o, float odx, float ody, float odz, float dt) {
    o->update_time += dt; /* :634-635 */
    o->update_counter += 1;
    for (size_t i = 0, n = (size_t)o->np * o->capp; i < n; i++) o->pyr_lists[i * 3] &= 0; /* :638-642 */
    for (int v = 0; v < o->V; ++v) {
        for (int p = 0; p < o->slots; p++) {
            float* r = PART(o, v, p);
            if (r[0] > 0.1f && r[0] < 6.f) { /* :649 */
                r[0] = 1.f;
                if (o->cfg.static_model) {
                    r[1] = 0.f; r[2] = 0.f; r[3] = 0.f; /* dsp_static.h:640-642: the static motion model */
                } else if (fabs(r[1] * r[2] * r[3]) < 1e-6) { /* :653 */
                } else {
                    r[1] += draw_v(o);
                    r[2] += draw_v(o);
                    r[3] += draw_v(o);
                }
                r[3] = 0.f; /* :662 */
                r[4] += dt * r[1] + odx; /* :665-667 */
                r[5] += dt * r[2] + ody;
                r[6] += dt * r[3] + odz;
                int nv;
                if (dspo_voxel_index(o, r[4], r[5], r[6], &nv)) {
                    (void)move_particle(o, nv, v, p, r); /* :674-682 */
                } else {
                    r[0] = 0.f; /* removeParticle :688,1277 */
                }
            }
        }
    }
}

/* ------------------------------------------------------------------ update */
/* mapUpdate :704-793, pass 1 (:709-735) without the final += (lambda + kappa) of :737 */
static void map_update_pass1(dsp_oracle* o, int add_const) {
    for (int i = 0; i < o->np; ++i) { /* pass 1, :709-739 */
        for (int j = 0; j < o->obs_count[i]; ++j) {
            float* ob = OBS(o, i, j);
            const int* nb = o->neighbors + i * o->nbs;
            for (int n = 0; n < nb[0]; ++n) {
                int b = nb[n + 1];
                for (int s = 0; s < o->capp; ++s) {
                    const int* e = PYR(o, b, s);
                    if (e[0] & 1) {
                        const float* r = PART(o, e[1], e[2]);
                        float gk = dspo_query_normal_pdf(o, r[4], ob[0], o->sigma_ob) *
                                   dspo_query_normal_pdf(o, r[5], ob[1], o->sigma_ob) *
                                   dspo_query_normal_pdf(o, r[6], ob[2], o->sigma_ob);
                        ob[3] += o->P_detection * r[7] * gk; /* :732 */
                    }
                }
            }
            if (add_const) ob[3] += (o->expected_new_born_objects + o->kappa); /* :737 */
        }
    }
}
/* mapUpdate pass 2 (:743-790) */
static void map_update_pass2(dsp_oracle* o) {
    for (int i = 0; i < o->np; i++) { /* pass 2, :743-790 */
        for (int s = 0; s < o->capp; s++) {
            const int* e = PYR(o, i, s);
            if (e[0] & 1) {
                const int* nb = o->neighbors + i * o->nbs;
                float* r = PART(o, e[1], e[2]);
                float px = r[4], py = r[5], pz = r[6];
                float dist = sqrtf(px * px + py * py + pz * pz);
                if (o->obs_maxlen[i] > 0.f && dist > o->obs_maxlen[i] + o->occl_margin) continue; /* :761, obstacle_thickness :70 */
                float sum = 0.f;
                for (int n = 0; n < nb[0]; ++n) {
                    int b = nb[n + 1];
                    for (int z = 0; z < o->obs_count[b]; ++z) {
                        float* ob = OBS(o, b, z);
                        float gk = dspo_query_normal_pdf(o, px, ob[0], o->sigma_ob) *
                                   dspo_query_normal_pdf(o, py, ob[1], o->sigma_ob) *
                                   dspo_query_normal_pdf(o, pz, ob[2], o->sigma_ob);
                        sum += o->P_detection * gk / ob[3]; /* :779 */
                    }
                }
                r[7] *= ((1 - o->P_detection) + sum); /* :786 */
                r[8] = o->update_time;
            }
        }
    }
}
/* mapUpdate :704-793 */
void dspo_map_update(dsp_oracle* o) {
    map_update_pass1(o, 1);
    map_update_pass2(o);
}
/* The two halves of mapUpdate, callable separately (test infrastructure for the Z-slab sharding
 * test: the per-observation sums of the slabs are added between them).  ck: pass 1 without the
 * constant of :737; weights: adds that constant, then pass 2. */
void dspo_map_update_ck(dsp_oracle* o) { map_update_pass1(o, 0); }
void dspo_map_update_weights(dsp_oracle* o) {
    for (int i = 0; i < o->np; ++i)
        for (int j = 0; j < o->obs_count[i]; ++j) OBS(o, i, j)[3] += (o->expected_new_born_objects + o->kappa);
    map_update_pass2(o);
}

/* Dempster-Shafer static/dynamic split of one birth source (:822-866).  Returns 0 if the source
 * lies outside the map (:827,847). */
static int birth_nstatic(dsp_oracle* o, float cx, float cy, float cz, int* out) {
    int pv;
    float ws = 0.f, wd = 0.f, wsd = 0.f;
    if (dspo_voxel_index(o, cx, cy, cz, &pv)) { /* :827 */
        for (int kk = 0; kk < o->slots; ++kk) {
            const float* r = PART(o, pv, kk);
            if (r[0] > 0.9f && r[0] < 14.f) { /* :830 */
                float v_abs = fabsf(r[1]) + fabsf(r[2]) + fabsf(r[3]);
                if (v_abs < 0.1f) ws += r[7];
                else if (v_abs < 0.5f) wsd += r[7];
                else wd += r[7];
            }
        }
    } else {
        return 0;
    }
    /* Dempster-Shafer :850-866 */
    float total = ws + wd + wsd;
    float m_s = ws / total, m_d = wd / total, m_sd = wsd / total;
    float p_s = (m_s + m_s + m_sd) * 0.5f;
    float p_d = (m_d + m_d + m_sd) * 0.5f;
    float p_s_n = p_s / (p_s + p_d);
    float fstat = (float)o->model_generated_nb * p_s_n;
    /* (int) of NaN when the voxel is empty: cvttss2si gives INT_MIN on x86,
     * so max(min_static, .) picks min_static. */
    int n_static = (fstat != fstat) ? INT_MIN : (int)fstat;
    if (n_static < o->min_static_nb) n_static = o->min_static_nb; /* :866 */
    *out = n_static;
    return 1;
}

static void birth_statics_init(dsp_oracle* o) {
    if (!o->nb_statics_init) { /* function statics frozen at first call :808-811 */
        const int n_nb = o->new_born_particle_number_each_point;
        o->min_static_nb = (int)((float)n_nb * 0.15f);
        o->model_generated_nb = (int)((float)n_nb * 0.8f);
        o->nb_statics_init = 1;
    }
}
/* test infrastructure for the Z-slab sharding test: n_static of every birth source as this
 * instance sees it (0 for sources outside the map), and an override used by dspo_add_newborn */
void dspo_compute_nstatic(dsp_oracle* o, int* out) {
    birth_statics_init(o);
    for (int q = 0; q < o->birth_n; q++) {
        const dspo_vpoint* pt = &o->birth[q];
        int ns = 0;
        if (!birth_nstatic(o, pt->x - o->current_position[0], pt->y - o->current_position[1],
                           pt->z - o->current_position[2], &ns)) ns = 0;
        out[q] = ns;
    }
}
void dspo_set_nstatic_override(dsp_oracle* o, const int* arr) { o->nstatic_override = arr; }

/* ------------------------------------------------------------------- birth */
/* mapAddNewBornParticlesByObservation :796-921 */
void dspo_add_newborn(dsp_oracle* o) {
    float norm = 0.f; /* :799-805 */
    for (int i = 0; i < o->np; i++)
        for (int j = 0; j < o->obs_count[i]; j++) norm += 1.f / OBS(o, i, j)[3];
    float w_new = o->new_born_particle_weight * norm;
    const int n_nb = o->new_born_particle_number_each_point;
    birth_statics_init(o);
    for (int q = 0; q < o->birth_n; q++) { /* :815 */
        const dspo_vpoint* pt = &o->birth[q];
        float cx = pt->x - o->current_position[0]; /* :818-820 */
        float cy = pt->y - o->current_position[1];
        float cz = pt->z - o->current_position[2];
        int n_static;
        if (o->cfg.static_model) {
            /* dsp_static.h:797-825: no voxel lookup for the source point (a source outside the map still draws and may
             * place children inside it) and no static/dynamic split: every child has zero velocity (:813-815) */
            n_static = n_nb;
        } else {
            if (!birth_nstatic(o, cx, cy, cz, &n_static)) continue; /* :847 */
            if (o->nstatic_override) n_static = o->nstatic_override[q];
        }
        for (int p = 0; p < n_nb; p++) { /* :868 */
            float px = cx + draw_p(o);
            float py = cy + draw_p(o);
            float pz = cz + draw_p(o);
            int idx;
            if (dspo_voxel_index(o, px, py, pz, &idx)) { /* :875 */
                float vx, vy, vz;
                if (p < n_static) {
                    vx = vy = vz = 0.f;
                } else if (pt->nx > -100.f && p < o->model_generated_nb) { /* :881 */
                    if (pt->intensity > 0.01f) {
                        vx = pt->nx + 4 * draw_v(o);
                        vy = pt->ny + 4 * draw_v(o);
                        vz = pt->nz + 4 * draw_v(o);
                    } else {
                        vx = vy = vz = 0.f;
                    }
                } else {
                    if (pt->intensity > 0.01f) { /* :894-897 */
                        vx = dspo_generate_random_float(o, -1.5f, 1.5f);
                        vy = dspo_generate_random_float(o, -1.5f, 1.5f);
                        vz = dspo_generate_random_float(o, -0.5f, 0.5f);
                    } else {
                        vx = vy = vz = 0.f;
                    }
                }
                vz = 0.f; /* :905-907 */
                add_a_particle(o, px, py, pz, vx, vy, vz, w_new, idx); /* :911 */
            }
        }
    }
}

/* ---------------------------------------------- occupancy, rollout, resample */
/* mapOccupancyCalculationAndResample :924-1057 */
void dspo_occupancy_resample(dsp_oracle* o) {
    const int M = o->cfg.max_particle_num_voxel, T = o->cfg.prediction_times;
    for (int v = 0; v < o->V; ++v) {
        float wsum = 0.f, vxs = 0.f, vys = 0.f, vzs = 0.f;
        int n = 0, n_old = 0;
        for (int p = 0; p < o->slots; p++) {
            float* r = PART(o, v, p);
            if (r[0] > 0.1f) {
                if (r[7] < 1e-3) { /* :941 (double compare, as written) */
                    r[0] = 0.f;
                } else {
                    if (r[0] < 10.f) { /* :944 */
                        ++n_old;
                        vxs += r[1]; vys += r[2]; vzs += r[3];
                        for (int t = 0; t < T; ++t) { /* :952-963 */
                            float pt = o->cfg.prediction_future_time[t];
                            float fx = r[4] + r[1] * pt;
                            float fy = r[5] + r[2] * pt;
                            float fz = r[6] + r[3] * pt;
                            int fi;
                            if (dspo_voxel_index(o, fx, fy, fz, &fi)) RES(o, fi)[4 + t] += r[7];
                        }
                    }
                    r[0] = 1.f; /* :968 */
                    ++n;
                    wsum += r[7];
                }
            }
        }
        float* out = RES(o, v);
        out[0] = wsum; /* :974 */
        if (n_old > 0) {
            out[1] = vxs / (float)n_old; out[2] = vys / (float)n_old; out[3] = vzs / (float)n_old;
        } else {
            out[1] = out[2] = out[3] = 0.f;
        }
        if (n < 5) continue; /* :986 */
        int n_after = n > M ? M : n; /* :992-997 */
        float w_after = wsum / (float)n_after;
        float acc_ori = 0.f, acc_new = w_after * 0.5f; /* :1005-1006 */
        for (int p = 0; p < o->slots; ++p) {
            float* r = PART(o, v, p);
            if (r[0] > 0.7f) { /* :1009 */
                acc_ori += r[7];
                if (acc_ori > acc_new) {
                    r[7] = w_after;
                    acc_new += w_after;
                    int full = 0, pi = 0;
                    while (acc_ori > acc_new) { /* :1021 */
                        int found = 0;
                        if (!full) {
                            for (; pi < o->slots; ++pi) {
                                float* d = PART(o, v, pi);
                                if (d[0] < 0.1f) {
                                    d[0] = 0.6f;
                                    for (int k = 1; k < 9; k++) d[k] = r[k];
                                    found = 1;
                                    break;
                                }
                            }
                        }
                        if (!found) {
                            r[7] += w_after; /* :1039 */
                            full = 1;
                        }
                        acc_new += w_after;
                    }
                } else {
                    r[0] = 0.f; /* :1048 */
                }
            }
        }
    }
}

/* ----------------------------------------------------------------- readout */
/* getOccupancyMap :385-402 */
int dspo_get_occupancy_map(dsp_oracle* o, float thr, float* xyz, int cap) {
    int n = 0;
    for (int i = 0; i < o->V; i++) {
        float* r = RES(o, i);
        if (r[0] > thr) {
            if (xyz && n < cap) dspo_voxel_center(o, i, &xyz[3 * n], &xyz[3 * n + 1], &xyz[3 * n + 2]);
            ++n;
        }
        for (int j = 4; j < o->rdim; ++j) r[j] = 0.f;
    }
    return n;
}
/* getOccupancyMapWithFutureStatus :405-426 */
int dspo_get_occupancy_map_with_future(dsp_oracle* o, float thr, float* xyz, int cap, float* fut) {
    const int T = o->cfg.prediction_times;
    int n = 0;
    for (int i = 0; i < o->V; i++) {
        float* r = RES(o, i);
        if (r[0] > thr) {
            if (xyz && n < cap) dspo_voxel_center(o, i, &xyz[3 * n], &xyz[3 * n + 1], &xyz[3 * n + 2]);
            ++n;
        }
        for (int t = 0; t < T; ++t) fut[(size_t)i * T + t] = r[t + 4];
        for (int j = 4; j < o->rdim; ++j) r[j] = 0.f;
    }
    return n;
}
/* clearOccupancyMapPrediction :431-438 */
void dspo_clear_future(dsp_oracle* o) {
    for (int i = 0; i < o->V; i++)
        for (int j = 4; j < o->rdim; ++j) RES(o, i)[j] = 0.f;
}

/* ------------------------------------------------------- velocity estimator */
/* velocityEstimationThread :1377-1544.
 * Third-party pieces that are NOT under /root/reference (parity unpinned):
 *  - pcl::EuclideanClusterExtraction + search::KdTree (PCL version = whatever
 *    the ROS distro ships, readme.md:21-24).  Published algorithm (Rusu 2009):
 *    seed points in index order, grow by radius search, keep clusters whose
 *    size is within [min,max], return them sorted by size, largest first.
 *    Restated with a brute-force radius search (neighbours visited in
 *    ascending distance, as a sorted KdTree search returns them).
 *  - saebyn/munkres-cpp (unpinned HEAD, readme.md:27-35): minimum-cost
 *    assignment; restated as the O(n^3) Hungarian algorithm; the reference
 *    only reads which (row,col) cells are assigned (:1481). */
static float cluster_distance(const cluster_feature* a, const cluster_feature* b) { /* :1369-1374 */
    float sq = (a->cx - b->cx) * (a->cx - b->cx) + (a->cy - b->cy) * (a->cy - b->cy) +
               (a->cz - b->cz) * (a->cz - b->cz);
    return sqrtf(sq);
}

/* Hungarian (Kuhn-Munkres) for an n_r x n_c cost matrix, min cost; assign[r] = c or -1 */
static void hungarian(const float* cost, int nr, int nc, int* assign) {
    int n = nr > nc ? nr : nc;
    double* a = (double*)malloc(sizeof(double) * (size_t)(n + 1) * (n + 1));
    double big = 0;
    for (int i = 0; i < nr * nc; i++) if (cost[i] > big) big = cost[i];
    for (int i = 1; i <= n; i++)
        for (int j = 1; j <= n; j++)
            a[i * (n + 1) + j] = (i <= nr && j <= nc) ? cost[(i - 1) * nc + (j - 1)] : big;
    double* u = (double*)calloc(n + 1, sizeof(double));
    double* vv = (double*)calloc(n + 1, sizeof(double));
    int* p = (int*)calloc(n + 1, sizeof(int));
    int* way = (int*)calloc(n + 1, sizeof(int));
    double* minv = (double*)malloc(sizeof(double) * (n + 1));
    char* used = (char*)malloc(n + 1);
    for (int i = 1; i <= n; i++) {
        p[0] = i;
        int j0 = 0;
        for (int j = 0; j <= n; j++) { minv[j] = 1e300; used[j] = 0; }
        do {
            used[j0] = 1;
            int i0 = p[j0], j1 = 0;
            double delta = 1e300;
            for (int j = 1; j <= n; j++)
                if (!used[j]) {
                    double cur = a[i0 * (n + 1) + j] - u[i0] - vv[j];
                    if (cur < minv[j]) { minv[j] = cur; way[j] = j0; }
                    if (minv[j] < delta) { delta = minv[j]; j1 = j; }
                }
            for (int j = 0; j <= n; j++)
                if (used[j]) { u[p[j]] += delta; vv[j] -= delta; }
                else minv[j] -= delta;
            j0 = j1;
        } while (p[j0] != 0);
        do {
            int j1 = way[j0];
            p[j0] = p[j1];
            j0 = j1;
        } while (j0);
    }
    for (int r = 0; r < nr; r++) assign[r] = -1;
    for (int j = 1; j <= n; j++)
        if (p[j] >= 1 && p[j] <= nr && j <= nc) assign[p[j] - 1] = j - 1;
    free(a); free(u); free(vv); free(p); free(way); free(minv); free(used);
}

typedef struct { float d; int i; } dist_idx;
static int cmp_dist(const void* a, const void* b) {
    float da = ((const dist_idx*)a)->d, db = ((const dist_idx*)b)->d;
    if (da < db) return -1;
    if (da > db) return 1;
    return ((const dist_idx*)a)->i - ((const dist_idx*)b)->i;
}
static int cmp_int_asc(const void* a, const void* b) { return *(const int*)a - *(const int*)b; }
typedef struct { int start, size; } cluster_span;
/* PCL sorts the clusters by size, largest first (std::sort: the order of equal sizes is unspecified there); equal
 * sizes keep their seed order here */
static int cmp_cluster_desc(const void* a, const void* b) {
    const int d = ((const cluster_span*)b)->size - ((const cluster_span*)a)->size;
    return d ? d : ((const cluster_span*)a)->start - ((const cluster_span*)b)->start;
}

/* pcl::EuclideanClusterExtraction (see the note above): clusters of pts[n][3] under the squared tolerance tol2 whose size is
 * within [min_sz, max_sz], largest first; order[] receives the members of the clusters one after the other (ascending index
 * inside a cluster), spans[] (room for n / min_sz + 1) where each cluster sits in order[].  Returns the number of clusters. */
static int euclidean_clusters(const float* ng, int n_ng, float tol2, int min_sz, int max_sz, int* order, cluster_span* spans) {
    char* processed = (char*)calloc(n_ng > 0 ? n_ng : 1, 1);
    int n_order = 0, n_spans = 0;
    int* queue = (int*)malloc(sizeof(int) * (n_ng > 0 ? n_ng : 1));
    dist_idx* nbrs = (dist_idx*)malloc(sizeof(dist_idx) * (n_ng > 0 ? n_ng : 1));
    for (int i = 0; i < n_ng; i++) {
        if (processed[i]) continue;
        int qn = 0, qi = 0;
        queue[qn++] = i; processed[i] = 1;
        while (qi < qn) {
            int c = queue[qi++];
            int nn = 0;
            for (int j = 0; j < n_ng; j++) {
                float dx = ng[3 * j] - ng[3 * c], dy = ng[3 * j + 1] - ng[3 * c + 1], dz = ng[3 * j + 2] - ng[3 * c + 2];
                float d2 = dx * dx + dy * dy + dz * dz;
                if (d2 <= tol2) { nbrs[nn].d = d2; nbrs[nn].i = j; nn++; }
            }
            qsort(nbrs, nn, sizeof(dist_idx), cmp_dist);
            for (int k = 0; k < nn; k++)
                if (!processed[nbrs[k].i]) { processed[nbrs[k].i] = 1; queue[qn++] = nbrs[k].i; }
        }
        if (qn >= min_sz && qn <= max_sz) {
            /* PCL's extractEuclideanClusters sorts (and uniques) the indices of every cluster before it returns them
             * (pcl/segmentation/impl/extract_clusters.hpp: std::sort(r.indices.begin(), r.indices.end())), so the points
             * of a cluster reach :1424-1429 and :1509-1519 in ascending index order, not in region-growing order */
            qsort(queue, qn, sizeof(int), cmp_int_asc);
            spans[n_spans].start = n_order; spans[n_spans].size = qn; n_spans++;
            memcpy(order + n_order, queue, sizeof(int) * qn);
            n_order += qn;
        }
    }
    qsort(spans, n_spans, sizeof(cluster_span), cmp_cluster_desc);
    free(processed); free(queue); free(nbrs);
    return n_spans;
}

/* test hooks for the two third-party algorithms (tests/test_oracle_kat.py checks them against independent implementations:
 * scipy's linear_sum_assignment, connected components of the radius graph) */
void dspo_hungarian(const float* cost, int nr, int nc, int* assign) { hungarian(cost, nr, nc, assign); }
int dspo_euclidean_clusters(const float* pts, int n, float tol, int min_sz, int max_sz, int* label) {
    int* order = (int*)malloc(sizeof(int) * (n > 0 ? n : 1));
    cluster_span* spans = (cluster_span*)malloc(sizeof(cluster_span) * (n / (min_sz > 0 ? min_sz : 1) + 1));
    const int ns = euclidean_clusters(pts, n, tol * tol, min_sz, max_sz, order, spans);
    for (int i = 0; i < n; i++) label[i] = -1;
    for (int c = 0; c < ns; c++)
        for (int k = 0; k < spans[c].size; k++) label[order[spans[c].start + k]] = c;   /* c = rank by size, largest first */
    free(order); free(spans);
    return ns;
}

void dspo_velocity_estimation(dsp_oracle* o) {
    if (o->cloud_view_n == 0) return; /* :1379: early return WITHOUT clearing the previous output */
    o->birth_n = 0;                   /* :1381 */
    int n_all = o->cloud_view_n;
    float* ng = (float*)malloc(sizeof(float) * 3 * n_all); int n_ng = 0; /* non_ground_points */
    float* st = (float*)malloc(sizeof(float) * 3 * n_all * 2); int n_st = 0; /* static_points */
    for (int i = 0; i < n_all; i++) { /* :1387-1398 */
        float x = o->cloud_view[3 * i] + o->current_position[0];
        float y = o->cloud_view[3 * i + 1] + o->current_position[1];
        float z = o->cloud_view[3 * i + 2] + o->current_position[2];
        if (z > o->voxel_filtered_resolution) { ng[3 * n_ng] = x; ng[3 * n_ng + 1] = y; ng[3 * n_ng + 2] = z; n_ng++; }
        else { st[3 * n_st] = x; st[3 * n_st + 1] = y; st[3 * n_st + 2] = z; n_st++; }
    }
    cluster_feature* dyn = NULL; int n_dyn = 0;
    birth_reserve(o, n_all);
    if (n_ng > 0) { /* :1406 */
        /* Euclidean clustering: tolerance 2*res_filter, size 5..10000 (:1411-1413) */
        float tol = 2 * o->voxel_filtered_resolution, tol2 = tol * tol;
        int* order = (int*)malloc(sizeof(int) * n_ng); /* concatenated cluster indices */
        cluster_span* spans = (cluster_span*)malloc(sizeof(cluster_span) * (n_ng / 5 + 1));
        const int n_spans = euclidean_clusters(ng, n_ng, tol2, 5, 10000, order, spans);
        char* possibly_dynamic = (char*)calloc(n_spans + 1, 1);
        dyn = (cluster_feature*)calloc(n_spans + 1, sizeof(cluster_feature));
        for (int c = 0; c < n_spans; c++) { /* :1419-1447 */
            cluster_feature f;
            memset(&f, 0, sizeof(f));
            f.match_seq = -1; f.vx = f.vy = f.vz = -10000.f;
            f.intensity = dspo_generate_random_float(o, 0.1f, 1.f); /* :1422 */
            for (int k = 0; k < spans[c].size; k++) {
                int id = order[spans[c].start + k];
                f.cx += ng[3 * id]; f.cy += ng[3 * id + 1]; f.cz += ng[3 * id + 2];
                ++f.point_num;
            }
            f.cx /= (float)f.point_num; f.cy /= (float)f.point_num; f.cz /= (float)f.point_num;
            if (spans[c].size > 200 || f.cz > 1.5) { /* DYNAMIC_CLUSTER_MAX_* :52-53,1436 */
                for (int k = 0; k < spans[c].size; k++) {
                    int id = order[spans[c].start + k];
                    st[3 * n_st] = ng[3 * id]; st[3 * n_st + 1] = ng[3 * id + 1]; st[3 * n_st + 2] = ng[3 * id + 2];
                    n_st++;
                }
                possibly_dynamic[c] = 0;
            } else {
                dyn[n_dyn++] = f;
                possibly_dynamic[c] = 1;
            }
        }
        const float distance_gate = 1.5f, maximum_velocity = 5.f; const int point_num_gate = 100; /* :1449-1451 */
        if (o->last_clusters_n > 0 && n_dyn > 0 && o->delt_t_from_last > 0.00001 && o->delt_t_from_last < 10.0) { /* :1454-1455 */
            int nl = o->last_clusters_n;
            float* cost = (float*)malloc(sizeof(float) * n_dyn * nl);
            float* gate = (float*)malloc(sizeof(float) * n_dyn * nl);
            int* assign = (int*)malloc(sizeof(int) * n_dyn);
            for (int r = 0; r < n_dyn; ++r)
                for (int c = 0; c < nl; ++c) { /* :1459-1472 */
                    float d = cluster_distance(&dyn[r], &o->last_clusters[c]);
                    if (abs(dyn[r].point_num - o->last_clusters[c].point_num) > point_num_gate || d >= distance_gate) {
                        gate[r * nl + c] = 0.f; cost[r * nl + c] = distance_gate * 5000.f;
                    } else {
                        gate[r * nl + c] = 1.f; cost[r * nl + c] = d / distance_gate * 1000.f;
                    }
                }
            hungarian(cost, n_dyn, nl, assign); /* :1474-1475 */
            for (int r = 0; r < n_dyn; ++r) { /* :1477-1499 */
                int c = assign[r];
                if (c >= 0 && gate[r * nl + c] > 0.01f) {
                    float dt = o->delt_t_from_last;
                    dyn[r].match_seq = c;
                    dyn[r].vx = (dyn[r].cx - o->last_clusters[c].cx) / dt;
                    dyn[r].vy = (dyn[r].cy - o->last_clusters[c].cy) / dt;
                    dyn[r].vz = (dyn[r].cz - o->last_clusters[c].cz) / dt;
                    dyn[r].v = sqrtf(dyn[r].vx * dyn[r].vx + dyn[r].vy * dyn[r].vy + dyn[r].vz * dyn[r].vz);
                    dyn[r].intensity = o->last_clusters[c].intensity;
                    if (dyn[r].v > maximum_velocity) { dyn[r].v = 0.f; dyn[r].vx = dyn[r].vy = dyn[r].vz = 0.f; }
                }
            }
            free(cost); free(gate); free(assign);
        }
        int dseq = 0; /* :1505-1524 */
        for (int c = 0; c < n_spans; c++) {
            if (possibly_dynamic[c]) {
                for (int k = 0; k < spans[c].size; k++) {
                    int id = order[spans[c].start + k];
                    dspo_vpoint* q = &o->birth[o->birth_n++];
                    q->x = ng[3 * id]; q->y = ng[3 * id + 1]; q->z = ng[3 * id + 2];
                    q->nx = dyn[dseq].vx; q->ny = dyn[dseq].vy; q->nz = dyn[dseq].vz;
                    q->intensity = dyn[dseq].intensity;
                }
                ++dseq;
            }
        }
        free(order); free(spans); free(possibly_dynamic);
    }
    birth_reserve(o, o->birth_n + n_st);
    for (int i = 0; i < n_st; i++) { /* :1529-1540 */
        dspo_vpoint* q = &o->birth[o->birth_n++];
        q->x = st[3 * i]; q->y = st[3 * i + 1]; q->z = st[3 * i + 2];
        q->nx = q->ny = q->nz = 0.f; q->intensity = 0.f;
    }
    free(o->last_clusters); /* :1542 */
    o->last_clusters = dyn;
    o->last_clusters_n = n_dyn;
    free(ng); free(st);
}

/* Test convenience (no counterpart in the reference): tag every in-FOV point as
 * a static birth source, in view order -- i.e. the velocity estimator's output
 * format (:1529-1540) without its ground/cluster re-ordering.  This is what
 * libdspmap_hip does when no birth cloud is supplied. */
void dspo_static_birth_cloud(dsp_oracle* o) {
    /* dsp_static.h:1285-1290 (and dsp_dynamic.h:1379-1381): an empty view returns BEFORE the previous output is
     * cleared, so the last non-empty frame's cloud is reused by the birth stage */
    if (o->cloud_view_n == 0) return;
    birth_reserve(o, o->cloud_view_n);
    o->birth_n = 0;
    for (int i = 0; i < o->cloud_view_n; i++) {
        dspo_vpoint* q = &o->birth[o->birth_n++];
        q->x = o->cloud_view[3 * i] + o->current_position[0];
        q->y = o->cloud_view[3 * i + 1] + o->current_position[1];
        q->z = o->cloud_view[3 * i + 2] + o->current_position[2];
        q->nx = q->ny = q->nz = 0.f; q->intensity = 0.f;
    }
}

/* ------------------------------------------------------------- whole frame */
/* DSPMap::update :181-353 (CSV dump :326-350 omitted: write-only debug aid) */
int dspo_update(dsp_oracle* o, int n_pts, int stride, const float* pts, float sx, float sy, float sz,
                double stamp, float qw, float qx, float qy, float qz) {
    if (!o->have_last) { /* function statics initialise to the first inputs :187-190 */
        o->last_px = sx; o->last_py = sy; o->last_pz = sz; o->last_stamp = stamp;
        o->have_last = 1;
    }
    if (fabs(qw) > 1.001f || fabs(qx) > 1.001f || fabs(qy) > 1.001f || fabs(qz) > 1.001f) return 0; /* :193-196 */
    float dx = sx - o->last_px, dy = sy - o->last_py, dz = sz - o->last_pz;
    float dt = (float)(stamp - o->last_stamp);
    if (fabs(dx) > 10.f || fabs(dy) > 10.f || fabs(dz) > 10.f || dt < 0.f || dt > 10.f) return 0; /* :203-208 */
    o->current_position[0] = o->last_px = sx; /* :213-216 */
    o->current_position[1] = o->last_py = sy;
    o->current_position[2] = o->last_pz = sz;
    o->last_stamp = stamp;
    o->delt_t_from_last = dt;
    dspo_bin_points(o, n_pts, stride, pts, qw, qx, qy, qz);
    /* the reference forks velocityEstimationThread here (:297) and joins at :311;
     * it shares no state with prediction/update, so running it first is equivalent */
    /* dsp_static.h has no velocity estimation: every in-FOV point is a zero-velocity birth source (:797-825) */
    if (o->cfg.static_model) dspo_static_birth_cloud(o);
    else if (o->use_vel_est == 1) dspo_velocity_estimation(o);
    else if (o->use_vel_est == 2) dspo_static_birth_cloud(o);
    dspo_map_prediction(o, -dx, -dy, -dz, dt); /* :300 */
    if (n_pts >= 0) dspo_map_update(o);        /* :303-307 */
    if (n_pts >= 0) dspo_add_newborn(o);       /* :314-316 */
    dspo_occupancy_resample(o);                /* :322 */
    return 1;
}

/* ==========================================================================
 * Caller-side pre-processing, src/map_sim_example.cpp:309-336.
 *
 * pcl::VoxelGrid is third-party code that is not under /root/reference (PCL, version unpinned by the
 * reference: readme.md:21-24).  Its published algorithm (pcl/filters/impl/voxel_grid.hpp; the arithmetic
 * below is the same in PCL 1.8 ... 1.12) is restated: bounding box of the finite points -> leaf lattice ->
 * (leaf index, point) pairs -> sort by leaf index -> one centroid per leaf in ascending leaf order.
 * std::sort leaves the order of equal keys unspecified; this restatement breaks ties by point index.
 * PARITY UNPINNED for this function: PCL is absent from the image and the reference has no fixture for it.
 * ========================================================================== */
typedef struct { int idx; int pt; } dspo_leaf_pair;
static int dspo_leaf_cmp(const void* a, const void* b) {
    const dspo_leaf_pair* x = (const dspo_leaf_pair*)a;
    const dspo_leaf_pair* y = (const dspo_leaf_pair*)b;
    if (x->idx != y->idx) return x->idx < y->idx ? -1 : 1;
    return x->pt < y->pt ? -1 : (x->pt > y->pt ? 1 : 0);
}

int dspo_preprocess_cloud(int n, const float* pts, int stride, float leaf, int swap_axes, float hx, float hy, float hz,
                          int max_points, float* out, int* n_leaves) {
    if (n_leaves) *n_leaves = 0;
    if (n <= 0) return 0;
    /* getMinMax3D over the finite points */
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    int nfin = 0;
    for (int i = 0; i < n; i++) {
        const float* p = pts + (size_t)i * stride;
        if (!isfinite(p[0]) || !isfinite(p[1]) || !isfinite(p[2])) continue;
        for (int a = 0; a < 3; a++) { if (p[a] < mn[a]) mn[a] = p[a]; if (p[a] > mx[a]) mx[a] = p[a]; }
        nfin++;
    }
    if (!nfin) return 0;
    const float inv = 1.0f / leaf;                     /* inverse_leaf_size_ */
    int min_b[3], div_b[3];
    for (int a = 0; a < 3; a++) {
        min_b[a] = (int)floorf(mn[a] * inv);
        div_b[a] = (int)floorf(mx[a] * inv) - min_b[a] + 1;
    }
    const int mul[3] = {1, div_b[0], div_b[0] * div_b[1]};   /* divb_mul_ */
    dspo_leaf_pair* iv = (dspo_leaf_pair*)malloc(sizeof(dspo_leaf_pair) * (size_t)nfin);
    int m = 0;
    for (int i = 0; i < n; i++) {
        const float* p = pts + (size_t)i * stride;
        if (!isfinite(p[0]) || !isfinite(p[1]) || !isfinite(p[2])) continue;
        const int ijk0 = (int)(floorf(p[0] * inv) - (float)min_b[0]);
        const int ijk1 = (int)(floorf(p[1] * inv) - (float)min_b[1]);
        const int ijk2 = (int)(floorf(p[2] * inv) - (float)min_b[2]);
        iv[m].idx = ijk0 * mul[0] + ijk1 * mul[1] + ijk2 * mul[2];
        iv[m].pt = i;
        m++;
    }
    qsort(iv, (size_t)m, sizeof(dspo_leaf_pair), dspo_leaf_cmp);
    int written = 0, leaves = 0;
    int first = 0;
    while (first < m) {
        int last = first + 1;
        while (last < m && iv[last].idx == iv[first].idx) last++;
        float c[3] = {0.f, 0.f, 0.f};                   /* centroid += point; centroid /= count */
        for (int k = first; k < last; k++) {
            const float* p = pts + (size_t)iv[k].pt * stride;
            c[0] += p[0]; c[1] += p[1]; c[2] += p[2];
        }
        const float cnt = (float)(last - first);
        c[0] /= cnt; c[1] /= cnt; c[2] /= cnt;
        leaves++;
        /* cloudCallback :319-336: axis swap, open-interval crop (inRange :190-197), stop when the buffer is full */
        if (written < max_points) {
            float x, y, z;
            if (swap_axes) { x = c[2]; y = -c[0]; z = -c[1]; } else { x = c[0]; y = c[1]; z = c[2]; }
            if (x > -hx && x < hx && y > -hy && y < hy && z > -hz && z < hz) {
                out[3 * written] = x; out[3 * written + 1] = y; out[3 * written + 2] = z;
                written++;
            }
        }
        first = last;
    }
    free(iv);
    if (n_leaves) *n_leaves = leaves;
    return written;
}
