/*
 * pcb_oracle.c -- CPU restatement of the reference's point-set operators.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity checker for the HIP
 * kernels in pointcloud-bridge_amd/csrc.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it; the product path never does.
 *
 * Parity status: PINNED.  Every function below is checked against outputs of
 * the reference itself (imported from /root/reference in the build container,
 * see tests/golden/make_golden.py) that are committed under tests/golden/.
 *
 * The reference implements these operators as ATen compositions in Python
 * (Highway_bridge/models/pointnet2_utils.py, Highway_bridge/models/DGCNN.py).
 * What is restated here is the arithmetic, in the exact fp32 operation order
 * the reference's CPU path produces:
 *
 *   square_distance  pointnet2_utils.py:7-14   d = ((-2*dot) + |s|^2) + |t|^2
 *                    with dot = fma(s2,t2, fma(s1,t1, s0*t0))   (K=3 sgemm)
 *                    and  |p|^2 = (x*x + y*y) + z*z              (no fma)
 *   farthest_point_sample   pointnet2_utils.py:63-80
 *   query_ball_point        pointnet2_utils.py:97-112
 *   three_nn (sort, first k) pointnet2_utils.py:185-188 (k=3), :253-256 (k=4)
 *   DGCNN.knn               DGCNN.py:49-70
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC (see oracle/Makefile).
 * -ffp-contract=off matters: every fma below is explicit, none is implied.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline float sq_norm3(const float *p)
{
    /* torch.sum(p ** 2, -1): (x*x + y*y) + z*z, each step rounded */
    float a = p[0] * p[0];
    float b = p[1] * p[1];
    float c = p[2] * p[2];
    float s = a + b;
    return s + c;
}

static inline float dot3_gemm(const float *s, const float *t)
{
    /* K=3 sgemm inner product as produced by the reference's torch.matmul on CPU */
    float d = s[0] * t[0];
    d = fmaf(s[1], t[1], d);
    d = fmaf(s[2], t[2], d);
    return d;
}

static inline float sqdist_expand(const float *s, float s2, const float *t, float t2)
{
    /* pointnet2_utils.py:11-13: dist = -2*matmul; dist += |src|^2; dist += |dst|^2 */
    float d = -2.0f * dot3_gemm(s, t);
    d = d + s2;
    d = d + t2;
    return d;
}

/* square_distance(src[B,N,3], dst[B,M,3]) -> out[B,N,M]   (pointnet2_utils.py:7-14) */
void orc_square_distance(const float *src, const float *dst, int B, int N, int M, float *out)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int i = 0; i < N; ++i) {
            const float *s = src + ((size_t)b * N + i) * 3;
            float s2 = sq_norm3(s);
            float *o = out + ((size_t)b * N + i) * M;
            for (int j = 0; j < M; ++j) {
                const float *t = dst + ((size_t)b * M + j) * 3;
                o[j] = sqdist_expand(s, s2, t, sq_norm3(t));
            }
        }
}

/*
 * farthest_point_sample(xyz[B,N,3], npoint) -> idx[B,npoint]   (pointnet2_utils.py:63-80)
 * start[b] is the index the reference draws with torch.randint on the CPU generator (:69);
 * the caller supplies it so the RNG contract stays in Python.
 * Per iteration (:73-78): record farthest; dist = ((dx*dx + dy*dy) + dz*dz);
 * running = dist where dist < running (strict); farthest = first argmax of running.
 */
void orc_fps(const float *xyz, int B, int N, int S, const int64_t *start, int64_t *out)
{
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; ++b) {
        const float *p = xyz + (size_t)b * N * 3;
        float *run = (float *)malloc(sizeof(float) * (size_t)N);
        for (int i = 0; i < N; ++i) run[i] = 1e10f;
        int64_t far = start[b];
        for (int s = 0; s < S; ++s) {
            out[(size_t)b * S + s] = far;
            const float cx = p[far * 3 + 0], cy = p[far * 3 + 1], cz = p[far * 3 + 2];
            float best = -INFINITY;
            int64_t besti = 0;
            for (int i = 0; i < N; ++i) {
                float dx = p[i * 3 + 0] - cx;
                float dy = p[i * 3 + 1] - cy;
                float dz = p[i * 3 + 2] - cz;
                float xx = dx * dx, yy = dy * dy, zz = dz * dz;
                float d = xx + yy;
                d = d + zz;
                if (d < run[i]) run[i] = d;
                if (run[i] > best) { best = run[i]; besti = i; }
            }
            far = besti;
        }
        free(run);
    }
}

/*
 * query_ball_point(radius, nsample, xyz[B,N,3], new_xyz[B,S,3]) -> idx[B,S,nsample]
 * (pointnet2_utils.py:97-112).  r2 is float32(radius**2): the reference compares the fp32
 * distance tensor against a Python scalar, which ATen converts to fp32.
 * Points with d > r2 are dropped (:105); survivors in ascending index order, first nsample
 * (:106); missing slots repeat the first survivor (:108-110).  With no survivor every slot
 * holds N, exactly as the reference leaves it.
 */
void orc_ball_query(const float *xyz, const float *new_xyz, int B, int N, int S,
                    float r2, int nsample, int64_t *out)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int s = 0; s < S; ++s) {
            const float *c = new_xyz + ((size_t)b * S + s) * 3;
            const float c2 = sq_norm3(c);
            int64_t *o = out + ((size_t)b * S + s) * nsample;
            int cnt = 0;
            for (int i = 0; i < N && cnt < nsample; ++i) {
                const float *t = xyz + ((size_t)b * N + i) * 3;
                float d = sqdist_expand(c, c2, t, sq_norm3(t));
                if (!(d > r2)) o[cnt++] = i;
            }
            int64_t first = cnt ? o[0] : (int64_t)N;
            for (int k = cnt; k < nsample; ++k) o[k] = first;
        }
}

/*
 * k nearest of xyz2 for every point of xyz1 (k = 3: FeaturePropagation
 * pointnet2_utils.py:185-188; k = 4: EnhancedFeaturePropagation :253-256).
 * d = square_distance(xyz1, xyz2) (src = xyz1), sorted ascending; equal distances keep
 * ascending index order (the reference's CPU sort is a stable merge sort).
 * Distances are returned unclamped: the expansion formula may give small negatives.
 */
void orc_three_nn(const float *xyz1, const float *xyz2, int B, int N, int S, int k,
                  float *out_d, int64_t *out_i)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int i = 0; i < N; ++i) {
            const float *s = xyz1 + ((size_t)b * N + i) * 3;
            const float s2 = sq_norm3(s);
            float bd[8];
            int64_t bi[8];
            int have = 0;
            for (int j = 0; j < S; ++j) {
                const float *t = xyz2 + ((size_t)b * S + j) * 3;
                float d = sqdist_expand(s, s2, t, sq_norm3(t));
                if (have < k || d < bd[have - 1]) {
                    int pos = have < k ? have : k - 1;
                    while (pos > 0 && d < bd[pos - 1]) {
                        bd[pos] = bd[pos - 1];
                        bi[pos] = bi[pos - 1];
                        --pos;
                    }
                    bd[pos] = d;
                    bi[pos] = j;
                    if (have < k) ++have;
                }
            }
            for (int q = 0; q < k; ++q) {
                out_d[((size_t)b * N + i) * k + q] = q < have ? bd[q] : INFINITY;
                out_i[((size_t)b * N + i) * k + q] = q < have ? bi[q] : 0;
            }
        }
}

/*
 * DGCNN.knn(x[B,D,N], k) -> idx[B,N,k]   (DGCNN.py:49-70).  x is given here already
 * transposed to [B,N,D] (DGCNN.py:60).  pd(i,j) = (|xi|^2 + (-2*<xi,xj>)) + |xj|^2 (:63-65);
 * the k largest of -pd, i.e. the k smallest pd, best first.  Ties: lowest index first
 * (torch.topk leaves tie order unspecified; parity tests compare tie-tolerantly).
 * <xi,xj> is an fmaf chain in channel order: bit for bit what the reference's sgemm (MKL 2024.2,
 * AVX-512) yields for K = 3 and K = 64 (tools/sgemm_order.py: 100 % of the inner products; 2-16 split
 * accumulators, reversed or unfused chains agree on 14-31 % only).
 * |x|^2 = torch.sum(x**2, dim=2) over the contiguous channel axis is ATen's vectorised reduction: for
 * D % 32 == 0 four 8-lane accumulators take the 8-channel chunks in turn (chunk t goes to accumulator
 * t % 4), the accumulators are added left to right, then the 8 lanes left to right (bit-identical for
 * D = 32, 64, 128, 256 -- the widths DGCNN's graphs have besides D = 3); other D: plain left-to-right
 * sum (pinned for D = 3 by the golden vectors, unpinned for the rest).
 * Also returns the distances so that exact ties can be told from mismatches.
 */
static float sumsq_aten(const float *p, int D)
{
    if (D % 32 != 0) {
        float s = 0.0f;
        for (int c = 0; c < D; ++c) {
            float sq = p[c] * p[c];
            s = c ? s + sq : sq;
        }
        return s;
    }
    float acc[4][8];
    for (int u = 0; u < 4; ++u)
        for (int l = 0; l < 8; ++l) {
            float a = p[u * 8 + l] * p[u * 8 + l];
            for (int c0 = u * 8 + 32; c0 < D; c0 += 32) {
                float sq = p[c0 + l] * p[c0 + l];
                a = a + sq;
            }
            acc[u][l] = a;
        }
    float r = 0.0f;
    for (int l = 0; l < 8; ++l) {
        float v = ((acc[0][l] + acc[1][l]) + acc[2][l]) + acc[3][l];
        r = l ? r + v : v;
    }
    return r;
}

void orc_knn(const float *x, int B, int N, int D, int k, int64_t *out_i, float *out_d)
{
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; ++b) {
        const float *xb = x + (size_t)b * N * D;
        float *nrm = (float *)malloc(sizeof(float) * (size_t)N);
        float *bd = (float *)malloc(sizeof(float) * (size_t)k);
        int64_t *bi = (int64_t *)malloc(sizeof(int64_t) * (size_t)k);
        for (int i = 0; i < N; ++i) nrm[i] = sumsq_aten(xb + (size_t)i * D, D);
        for (int i = 0; i < N; ++i) {
            const float *xi = xb + (size_t)i * D;
            int have = 0;
            for (int j = 0; j < N; ++j) {
                const float *xj = xb + (size_t)j * D;
                float dot = xi[0] * xj[0];
                for (int c = 1; c < D; ++c) dot = fmaf(xi[c], xj[c], dot);
                float d = nrm[i] + (-2.0f * dot);
                d = d + nrm[j];
                if (have < k || d < bd[have - 1]) {
                    int pos = have < k ? have : k - 1;
                    while (pos > 0 && d < bd[pos - 1]) {
                        bd[pos] = bd[pos - 1];
                        bi[pos] = bi[pos - 1];
                        --pos;
                    }
                    bd[pos] = d;
                    bi[pos] = j;
                    if (have < k) ++have;
                }
            }
            for (int q = 0; q < k; ++q) {
                out_i[((size_t)b * N + i) * k + q] = q < have ? bi[q] : 0;
                if (out_d) out_d[((size_t)b * N + i) * k + q] = q < have ? bd[q] : INFINITY;
            }
        }
        free(nrm);
        free(bd);
        free(bi);
    }
}

/*
 * index_points(points[B,N,C], idx[B,M]) -> out[B,M,C] with idx clamped to [0,N-1]
 * (pointnet2_utils.py:17-39; the clamp is :34-36).
 */
void orc_gather_rows(const float *points, const int64_t *idx, int B, int N, int C, int M, float *out)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int m = 0; m < M; ++m) {
            int64_t j = idx[(size_t)b * M + m];
            if (j < 0) j = 0;
            if (j > N - 1) j = N - 1;
            memcpy(out + ((size_t)b * M + m) * C, points + ((size_t)b * N + j) * C, sizeof(float) * (size_t)C);
        }
}

/*
 * Inverse-distance interpolation (pointnet2_utils.py:191-196 / :259-267):
 * w = 1/(d + 1e-8); w /= sum_k w; out[b,n,:] = sum_k w_k * feat[b, idx_k, :]  (k ascending).
 * feat is [B,S,C] (the reference transposes points2 to that layout first).
 */
void orc_interpolate(const float *feat, const float *d, const int64_t *idx,
                     int B, int N, int S, int C, int k, float *out, float *out_w)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int n = 0; n < N; ++n) {
            float w[8];
            float norm = 0.0f;
            for (int q = 0; q < k; ++q) {
                w[q] = 1.0f / (d[((size_t)b * N + n) * k + q] + 1e-8f);
                norm = q ? norm + w[q] : w[q];
            }
            for (int q = 0; q < k; ++q) {
                w[q] = w[q] / norm;
                if (out_w) out_w[((size_t)b * N + n) * k + q] = w[q];
            }
            float *o = out + ((size_t)b * N + n) * C;
            for (int c = 0; c < C; ++c) {
                float acc = 0.0f;
                for (int q = 0; q < k; ++q) {
                    int64_t j = idx[((size_t)b * N + n) * k + q];
                    if (j < 0) j = 0;
                    if (j > S - 1) j = S - 1;
                    float v = feat[((size_t)b * S + j) * C + c] * w[q];
                    acc = q ? acc + v : v;
                }
                o[c] = acc;
            }
        }
}
