// velocity_estimator.cpp -- see velocity_estimator.h.  Host C++ (compiled with
// -ffp-contract=off so the FOV test matches the device kernel bit for bit).
#include "velocity_estimator.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

static void quat_mul(const float a[4], const float b[4], float r[4]) {
    r[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    r[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    r[2] = a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3];
    r[3] = a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1];
}
static void rotate(const float* v, const float q[4], float out[3]) {  // dsp_dynamic.h:1303-1322
    const float vq[4] = {0.f, v[0], v[1], v[2]};
    const float n2 = q[1] * q[1] + q[2] * q[2] + q[3] * q[3] + q[0] * q[0];
    const float inv[4] = {q[0] / n2, -q[1] / n2, -q[2] / n2, -q[3] / n2};
    float t[4], r[4];
    quat_mul(q, vq, t);
    quat_mul(t, inv, r);
    out[0] = r[1]; out[1] = r[2]; out[2] = r[3];
}
static inline float dot3(const float* p, const float* n) { return p[0] * n[0] + p[1] * n[1] + p[2] * n[2]; }

void VelocityEstimator::configure(int half_fov_h, int half_fov_v, int A) {
    np_h_ = half_fov_h * 2 / A;
    np_v_ = half_fov_v * 2 / A;
    ph0_.assign((size_t)(np_h_ + 1) * 3, 0.f);
    pv0_.assign((size_t)(np_v_ + 1) * 3, 0.f);
    const float pi_f = 3.14159265358979323846f;
    const float ang = (float)A / 180.f * pi_f;
    const int he = half_fov_h / A, ve = half_fov_v / A;
    for (int i = -he; i <= he; i++) { ph0_[(i + he) * 3] = -sinf((float)i * ang); ph0_[(i + he) * 3 + 1] = cosf((float)i * ang); }
    for (int i = -ve; i <= ve; i++) { pv0_[(i + ve) * 3] = sinf((float)i * ang); pv0_[(i + ve) * 3 + 2] = cosf((float)i * ang); }
    ph_ = ph0_; pv_ = pv0_;
    configured_ = true;
}

void VelocityEstimator::rotate_and_filter(const float* pts, int n, const float q[4], std::vector<float>& view) {
    view.clear();
    // only the four outer planes are needed for ifInPyramidsArea (:1329-1339)
    float h0[3], hN[3], v0[3], vN[3];
    rotate(&ph0_[0], q, h0); rotate(&ph0_[(size_t)np_h_ * 3], q, hN);
    rotate(&pv0_[0], q, v0); rotate(&pv0_[(size_t)np_v_ * 3], q, vN);
    // rotate() with its per-call constants (|q|^2, the inverse quaternion) hoisted: same operations, same order
    const float n2 = q[1] * q[1] + q[2] * q[2] + q[3] * q[3] + q[0] * q[0];
    const float inv[4] = {q[0] / n2, -q[1] / n2, -q[2] / n2, -q[3] / n2};
    view.resize((size_t)n * 3);
    size_t k = 0;
    for (int i = 0; i < n; i++) {
        const float vq[4] = {0.f, pts[3 * (size_t)i], pts[3 * (size_t)i + 1], pts[3 * (size_t)i + 2]};
        float t[4], rr[4];
        quat_mul(q, vq, t);
        quat_mul(t, inv, rr);
        const float* r = rr + 1;
        if (dot3(r, h0) >= 0.f && dot3(r, hN) <= 0.f && dot3(r, v0) <= 0.f && dot3(r, vN) >= 0.f) {
            view[k] = r[0]; view[k + 1] = r[1]; view[k + 2] = r[2];
            k += 3;
        }
    }
    view.resize(k);
}

// Kuhn-Munkres, minimum cost, rectangular (padded); assign[r] = c or -1
static void hungarian(const std::vector<float>& cost, int nr, int nc, std::vector<int>& assign) {
    const int n = std::max(nr, nc);
    double big = 0;
    for (float c : cost) big = std::max(big, (double)c);
    std::vector<double> a((size_t)(n + 1) * (n + 1), big), u(n + 1, 0.0), v(n + 1, 0.0), minv(n + 1);
    std::vector<int> p(n + 1, 0), way(n + 1, 0);
    std::vector<char> used(n + 1);
    for (int i = 1; i <= nr; i++)
        for (int j = 1; j <= nc; j++) a[(size_t)i * (n + 1) + j] = cost[(size_t)(i - 1) * nc + (j - 1)];
    for (int i = 1; i <= n; i++) {
        p[0] = i;
        int j0 = 0;
        std::fill(minv.begin(), minv.end(), 1e300);
        std::fill(used.begin(), used.end(), 0);
        do {
            used[j0] = 1;
            const int i0 = p[j0];
            int j1 = 0;
            double delta = 1e300;
            for (int j = 1; j <= n; j++)
                if (!used[j]) {
                    const double cur = a[(size_t)i0 * (n + 1) + j] - u[i0] - v[j];
                    if (cur < minv[j]) { minv[j] = cur; way[j] = j0; }
                    if (minv[j] < delta) { delta = minv[j]; j1 = j; }
                }
            for (int j = 0; j <= n; j++)
                if (used[j]) { u[p[j]] += delta; v[j] -= delta; } else minv[j] -= delta;
            j0 = j1;
        } while (p[j0] != 0);
        do { const int j1 = way[j0]; p[j0] = p[j1]; j0 = j1; } while (j0);
    }
    assign.assign(nr, -1);
    for (int j = 1; j <= n; j++)
        if (p[j] >= 1 && p[j] <= nr && j <= nc) assign[p[j] - 1] = j - 1;
}

static float rand_float(float lo, float hi) {  // generateRandomFloat :1551-1553
    return lo + (float)rand() / ((float)(RAND_MAX / (hi - lo)));
}

void VelocityEstimator::run(const std::vector<float>& view, const float cur[3], float dt, float res_filter,
                            std::vector<dspmap_vpoint>& out) {
    const int n_all = (int)(view.size() / 3);
    if (n_all == 0) return;  // :1379 (previous output is kept)
    out.clear();             // :1381
    std::vector<float> ng, st;  // non_ground_points / static_points (world frame)
    ng.reserve(view.size()); st.reserve(view.size());
    for (int i = 0; i < n_all; i++) {  // :1387-1398
        const float x = view[3 * i] + cur[0], y = view[3 * i + 1] + cur[1], z = view[3 * i + 2] + cur[2];
        std::vector<float>& dst = (z > res_filter) ? ng : st;
        dst.push_back(x); dst.push_back(y); dst.push_back(z);
    }
    const int n_ng = (int)(ng.size() / 3);
    std::vector<Cluster> dyn;
    if (n_ng > 0) {
        // Euclidean cluster extraction, tolerance 2*res_filter, sizes 5..10000 (:1410-1417):
        // seeds in index order, breadth-first growth by radius search (hash grid, cell = tolerance),
        // clusters returned largest first.
        const float tol = 2 * res_filter, tol2 = tol * tol;
        // radius search = a uniform grid with cell = tolerance: counting sort of the points into the cells of their
        // bounding box (CSR), 27 cells per query.  (A cloud spread over more than 2^18 cells falls back to one cell per
        // axis slab by clamping -- still correct, the distance test decides.)
        auto cell = [&](float x) { return (long long)floorf(x / tol); };
        long long lo[3] = {cell(ng[0]), cell(ng[1]), cell(ng[2])}, hi[3] = {lo[0], lo[1], lo[2]};
        std::vector<int> cxyz((size_t)n_ng * 3);
        for (int i = 0; i < n_ng; i++)
            for (int a = 0; a < 3; a++) {
                const long long c = cell(ng[3 * i + a]);
                lo[a] = std::min(lo[a], c); hi[a] = std::max(hi[a], c);
            }
        long long dim[3];
        int shift[3] = {0, 0, 0};   // cells merged 2^shift to one along an axis if the box is huge (keeps the grid small)
        for (;;) {
            for (int a = 0; a < 3; a++) dim[a] = ((hi[a] - lo[a]) >> shift[a]) + 1;
            if (dim[0] * dim[1] * dim[2] <= (1ll << 18)) break;
            int big = 0;
            for (int a = 1; a < 3; a++) if (dim[a] > dim[big]) big = a;
            ++shift[big];
        }
        const size_t n_cells = (size_t)(dim[0] * dim[1] * dim[2]);
        cell_start_.assign(n_cells + 1, 0);
        std::vector<int> cid(n_ng);
        for (int i = 0; i < n_ng; i++) {
            for (int a = 0; a < 3; a++) cxyz[3 * i + a] = (int)((cell(ng[3 * i + a]) - lo[a]) >> shift[a]);
            cid[i] = (int)(((long long)cxyz[3 * i + 2] * dim[1] + cxyz[3 * i + 1]) * dim[0] + cxyz[3 * i]);
            ++cell_start_[cid[i] + 1];
        }
        for (size_t c = 0; c < n_cells; c++) cell_start_[c + 1] += cell_start_[c];
        // every cell keeps its UNPROCESSED points in [cell_start, cell_end): a point leaves its cell (swap with the
        // last active one) the moment it is queued, so a query only ever touches candidates that can still join
        std::vector<int> cell_pts(n_ng), cell_end(cell_start_.begin(), cell_start_.end() - 1), where(n_ng);
        for (int i = 0; i < n_ng; i++) { where[i] = cell_end[cid[i]]; cell_pts[cell_end[cid[i]]++] = i; }
        std::vector<char> processed(n_ng, 0);
        auto retire = [&](int j) {
            processed[j] = 1;
            const int last = --cell_end[cid[j]], moved = cell_pts[last];
            cell_pts[where[j]] = moved; where[moved] = where[j];
            cell_pts[last] = j; where[j] = last;
        };
        std::vector<std::vector<int>> clusters;
        std::vector<int> queue;
        std::vector<std::pair<float, int>> nbrs;
        for (int i = 0; i < n_ng; i++) {
            if (processed[i]) continue;
            queue.clear();
            queue.push_back(i);
            retire(i);
            for (size_t qi = 0; qi < queue.size(); ++qi) {
                const int c = queue[qi];
                const float cx = ng[3 * c], cy = ng[3 * c + 1], cz = ng[3 * c + 2];
                nbrs.clear();
                // a merged axis (shift > 0) holds >= 2 tolerance cells per grid cell: +-1 grid cell still covers +-tol
                for (int dz = -1; dz <= 1; dz++) {
                    const int z = cxyz[3 * c + 2] + dz;
                    if (z < 0 || z >= dim[2]) continue;
                    for (int dy = -1; dy <= 1; dy++) {
                        const int y = cxyz[3 * c + 1] + dy;
                        if (y < 0 || y >= dim[1]) continue;
                        const int x0 = std::max(cxyz[3 * c] - 1, 0), x1 = std::min(cxyz[3 * c] + 1, (int)dim[0] - 1);
                        const size_t row = ((size_t)z * dim[1] + y) * dim[0];
                        for (int x = x0; x <= x1; ++x)
                            for (int k = cell_start_[row + x]; k < cell_end[row + x]; ++k) {
                                const int j = cell_pts[k];
                                const float ex = ng[3 * j] - cx, ey = ng[3 * j + 1] - cy, ez = ng[3 * j + 2] - cz;
                                const float d2 = ex * ex + ey * ey + ez * ez;
                                if (d2 <= tol2) nbrs.emplace_back(d2, j);
                            }
                    }
                }
                for (auto& nb : nbrs) { retire(nb.second); queue.push_back(nb.second); }
            }
            if (queue.size() >= 5 && queue.size() <= 10000) {
                // PCL returns every cluster's indices sorted ascending (extract_clusters.hpp sorts r.indices): the cluster is the
                // connected component of the radius graph, the growth order leaves no trace
                std::sort(queue.begin(), queue.end());
                clusters.push_back(queue);
            }
        }
        std::stable_sort(clusters.begin(), clusters.end(),
                         [](const std::vector<int>& a, const std::vector<int>& b) { return a.size() > b.size(); });
        std::vector<char> possibly_dynamic(clusters.size(), 0);
        for (size_t c = 0; c < clusters.size(); c++) {  // :1419-1447
            Cluster f;
            f.intensity = rand_float(0.1f, 1.f);
            for (int id : clusters[c]) { f.cx += ng[3 * id]; f.cy += ng[3 * id + 1]; f.cz += ng[3 * id + 2]; ++f.point_num; }
            f.cx /= (float)f.point_num; f.cy /= (float)f.point_num; f.cz /= (float)f.point_num;
            if (clusters[c].size() > 200 || f.cz > 1.5) {  // DYNAMIC_CLUSTER_MAX_POINT_NUM / _CENTER_HEIGHT :52-53
                for (int id : clusters[c]) { st.push_back(ng[3 * id]); st.push_back(ng[3 * id + 1]); st.push_back(ng[3 * id + 2]); }
            } else {
                dyn.push_back(f);
                possibly_dynamic[c] = 1;
            }
        }
        const float distance_gate = 1.5f, maximum_velocity = 5.f;
        const int point_num_gate = 100;
        if (!last_.empty() && !dyn.empty() && dt > 0.00001 && dt < 10.0) {  // :1454-1455
            const int nr = (int)dyn.size(), nc = (int)last_.size();
            std::vector<float> cost((size_t)nr * nc), gate((size_t)nr * nc);
            for (int r = 0; r < nr; ++r)
                for (int c = 0; c < nc; ++c) {
                    const float ex = dyn[r].cx - last_[c].cx, ey = dyn[r].cy - last_[c].cy, ez = dyn[r].cz - last_[c].cz;
                    const float d = sqrtf(ex * ex + ey * ey + ez * ez);  // clusterDistance :1369-1374
                    if (abs(dyn[r].point_num - last_[c].point_num) > point_num_gate || d >= distance_gate) {
                        gate[(size_t)r * nc + c] = 0.f; cost[(size_t)r * nc + c] = distance_gate * 5000.f;
                    } else {
                        gate[(size_t)r * nc + c] = 1.f; cost[(size_t)r * nc + c] = d / distance_gate * 1000.f;
                    }
                }
            std::vector<int> assign;
            hungarian(cost, nr, nc, assign);
            for (int r = 0; r < nr; ++r) {  // :1477-1499
                const int c = assign[r];
                if (c >= 0 && gate[(size_t)r * nc + c] > 0.01f) {
                    dyn[r].vx = (dyn[r].cx - last_[c].cx) / dt;
                    dyn[r].vy = (dyn[r].cy - last_[c].cy) / dt;
                    dyn[r].vz = (dyn[r].cz - last_[c].cz) / dt;
                    dyn[r].v = sqrtf(dyn[r].vx * dyn[r].vx + dyn[r].vy * dyn[r].vy + dyn[r].vz * dyn[r].vz);
                    dyn[r].intensity = last_[c].intensity;
                    if (dyn[r].v > maximum_velocity) { dyn[r].v = 0.f; dyn[r].vx = dyn[r].vy = dyn[r].vz = 0.f; }
                }
            }
        }
        size_t dseq = 0;  // :1505-1524
        for (size_t c = 0; c < clusters.size(); c++) {
            if (!possibly_dynamic[c]) continue;
            for (int id : clusters[c]) {
                dspmap_vpoint p;
                p.x = ng[3 * id]; p.y = ng[3 * id + 1]; p.z = ng[3 * id + 2];
                p.nx = dyn[dseq].vx; p.ny = dyn[dseq].vy; p.nz = dyn[dseq].vz;
                p.intensity = dyn[dseq].intensity;
                out.push_back(p);
            }
            ++dseq;
        }
    }
    for (size_t i = 0; i < st.size() / 3; i++) {  // :1529-1540
        dspmap_vpoint p;
        p.x = st[3 * i]; p.y = st[3 * i + 1]; p.z = st[3 * i + 2];
        p.nx = p.ny = p.nz = 0.f; p.intensity = 0.f;
        out.push_back(p);
    }
    last_ = dyn;  // :1542
}

int VelocityEstimator::export_last(float* out5, int cap) const {
    const int n = std::min((int)last_.size(), cap);
    for (int i = 0; i < n; ++i) {
        out5[i * 5] = last_[i].cx; out5[i * 5 + 1] = last_[i].cy; out5[i * 5 + 2] = last_[i].cz;
        memcpy(&out5[i * 5 + 3], &last_[i].point_num, sizeof(int));
        out5[i * 5 + 4] = last_[i].intensity;
    }
    return n;
}
void VelocityEstimator::import_last(const float* in5, int n) {
    last_.assign((size_t)std::max(n, 0), Cluster());
    for (int i = 0; i < n; ++i) {
        last_[i].cx = in5[i * 5]; last_[i].cy = in5[i * 5 + 1]; last_[i].cz = in5[i * 5 + 2];
        memcpy(&last_[i].point_num, &in5[i * 5 + 3], sizeof(int));
        last_[i].intensity = in5[i * 5 + 4];
    }
}
