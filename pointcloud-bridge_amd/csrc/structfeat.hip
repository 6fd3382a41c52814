// Local-structure descriptor of a point's k nearest neighbours for gfx950.
//
// Replaces BridgeStructureEncoding.forward steps 3-5 and get_structure_features,
// models/attention_modules.py:595-603 and :620-687 of the reference: there a [B,N,k,3] gather, a
// batched 3x3 eigh, a [B*N,k,k] direction-similarity bmm (1 GiB at B=16, N=16384, k=32) and a dozen
// reductions.  Here one lane owns one point: its k <= 32 neighbour offsets live in registers, every
// statistic is a short loop over them, the symmetric 3x3 eigenvalues come from cyclic Jacobi
// rotations.  The offsets rel = x_j - x_i are optionally written out for the encoder's first 1x1
// convolution.  fp32 arithmetic as in the reference; the Jacobi sweeps run in fp64 so that the
// eigenvalues carry only the rounding of the fp32 covariance itself.
// HBM-bound gather: 12*k bytes of neighbour coordinates (L2 hits) + 8*k index bytes per point in,
// 52 (+12*k) bytes out.
#include "pcb_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kFeat = 13;

__device__ __forceinline__ void jacobi_rotate(double &app, double &aqq, double &apq, double &arp,
                                              double &arq)
{
    // annihilate a[p][q]; r is the third index.  Stable form of Rutishauser (tan of the angle).
    if (fabs(apq) < 1e-300) return;
    const double theta = (aqq - app) / (2.0 * apq);
    const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
    const double c = 1.0 / sqrt(t * t + 1.0);
    const double s = t * c;
    app -= t * apq;
    aqq += t * apq;
    apq = 0.0;
    const double rp = c * arp - s * arq;
    const double rq = s * arp + c * arq;
    arp = rp;
    arq = rq;
}

// ascending eigenvalues of the symmetric matrix [[a00,a01,a02],[a01,a11,a12],[a02,a12,a22]]
__device__ __forceinline__ void eigvals_sym3(double a00, double a01, double a02, double a11,
                                             double a12, double a22, float *e)
{
#pragma unroll 1
    for (int sweep = 0; sweep < 8; ++sweep) {
        const double off = fabs(a01) + fabs(a02) + fabs(a12);
        if (off <= 1e-18 * (fabs(a00) + fabs(a11) + fabs(a22))) break;
        jacobi_rotate(a00, a11, a01, a02, a12);  // (p,q) = (0,1), r = 2
        jacobi_rotate(a00, a22, a02, a01, a12);  // (0,2), r = 1: a[r][q] = a12
        jacobi_rotate(a11, a22, a12, a01, a02);  // (1,2), r = 0
    }
    double lo = a00, mid = a11, hi = a22, tmp;
    if (lo > mid) { tmp = lo; lo = mid; mid = tmp; }
    if (mid > hi) { tmp = mid; mid = hi; hi = tmp; }
    if (lo > mid) { tmp = lo; lo = mid; mid = tmp; }
    e[0] = (float)lo;
    e[1] = (float)mid;
    e[2] = (float)hi;
}

template <int KMAX>
__global__ __launch_bounds__(kThreads) void structure_features_kernel(
    const float *__restrict__ xyz, const int64_t *__restrict__ idx, int B, int N, int k,
    float *__restrict__ feat, float *__restrict__ rel_out)
{
    const long row = (long)blockIdx.x * kThreads + threadIdx.x;
    if (row >= (long)B * N) return;
    const int b = (int)(row / N);
    const float *__restrict__ p = xyz + (size_t)b * N * 3;
    const int64_t *__restrict__ nb = idx + (size_t)row * k;
    const float cx = xyz[row * 3 + 0], cy = xyz[row * 3 + 1], cz = xyz[row * 3 + 2];

    float rx[KMAX], ry[KMAX], rz[KMAX];
#pragma unroll
    for (int j = 0; j < KMAX; ++j) {
        const int jj = j < k ? j : 0;
        const int src = clamp_index(nb[jj], N);
        rx[j] = __fsub_rn(p[src * 3 + 0], cx);  // neighbors - center, :600
        ry[j] = __fsub_rn(p[src * 3 + 1], cy);
        rz[j] = __fsub_rn(p[src * 3 + 2], cz);
    }
    if (rel_out) {
        float *__restrict__ ro = rel_out + (size_t)row * k * 3;
#pragma unroll
        for (int j = 0; j < KMAX; ++j)
            if (j < k) {
                ro[j * 3 + 0] = rx[j];
                ro[j * 3 + 1] = ry[j];
                ro[j * 3 + 2] = rz[j];
            }
    }

    const float fk = (float)k, fk1 = (float)(k - 1);
    // second-moment matrix rel^T rel / (k-1) (:631, not centred) and the plain sums
    float sxx = 0, sxy = 0, sxz = 0, syy = 0, syz = 0, szz = 0, sx = 0, sy = 0, sz = 0;
    float nsx = 0, nsy = 0, nsz = 0, zmax = -INFINITY, zmin = INFINITY;
#pragma unroll
    for (int j = 0; j < KMAX; ++j)
        if (j < k) {
            const float x = rx[j], y = ry[j], z = rz[j];
            sxx += x * x; sxy += x * y; sxz += x * z;
            syy += y * y; syz += y * z; szz += z * z;
            sx += x; sy += y; sz += z;
            const float inv = 1.0f / (sqrtf(x * x + y * y + z * z) + 1e-8f);  // :657
            nsx += x * inv; nsy += y * inv; nsz += z * inv;
            zmax = fmaxf(zmax, z);
            zmin = fminf(zmin, z);
        }
    float e[3];
    eigvals_sym3((double)(sxx / fk1), (double)(sxy / fk1), (double)(sxz / fk1), (double)(syy / fk1),
                 (double)(syz / fk1), (double)(szz / fk1), e);
    const float den = e[0] + 1e-8f;  // the reference divides by the SMALLEST eigenvalue (:637-639)
    const float mx = sx / fk, my = sy / fk, mz = sz / fk;

    // distances to the neighbourhood's mean offset (:646-647), and per-axis spread
    float dmax = -INFINITY, dsum = 0, vx = 0, vy = 0, vz = 0;
    float dist[KMAX];
#pragma unroll
    for (int j = 0; j < KMAX; ++j)
        if (j < k) {
            const float x = rx[j] - mx, y = ry[j] - my, z = rz[j] - mz;
            const float d = sqrtf(x * x + y * y + z * z);
            dist[j] = d;
            dmax = fmaxf(dmax, d);
            dsum += d;
            vx += x * x; vy += y * y; vz += z * z;
        }
    const float dmean = dsum / fk;
    float dvar = 0;
#pragma unroll
    for (int j = 0; j < KMAX; ++j)
        if (j < k) {
            const float t = dist[j] - dmean;
            dvar += t * t;
        }
    const float stdx = sqrtf(vx / fk1), stdy = sqrtf(vy / fk1), stdz = sqrtf(vz / fk1);

    float *__restrict__ o = feat + (size_t)row * kFeat;
    o[0] = (e[0] - e[1]) / den;                                   // "linearity"  :637
    o[1] = (e[1] - e[2]) / den;                                   // "planarity"  :638
    o[2] = e[2] / den;                                            // "sphericity" :639
    o[3] = dmax;                                                  // local_radius :650
    o[4] = dmean;                                                 // mean_dist    :651
    o[5] = sqrtf(dvar / fk1);                                     // std_dist     :652 (unbiased)
    // mean over all (j,l) of n_j . n_l  ==  |sum_j n_j|^2 / k^2   (:658-662)
    o[6] = (nsx * nsx + nsy * nsy + nsz * nsz) / (fk * fk);
    o[7] = stdz;                                                  // z_variation  :666
    o[8] = zmax - zmin;                                           // z_range      :667
    o[9] = mx;                                                    // mean_rel_pos :671
    o[10] = my;
    o[11] = mz;
    o[12] = sqrtf(stdx * stdx + stdy * stdy + stdz * stdz);       // |std(rel)|   :680
}

}  // namespace

extern "C" int pcb_structure_features(const float *xyz, const int64_t *idx, int B, int N, int k,
                                      float *feat, float *rel, void *stream)
{
    if (!xyz || !idx || !feat || B <= 0 || N <= 0) return PCB_ERR_INVALID_ARG;
    if (k < 2 || k > 32) return PCB_ERR_UNSUPPORTED;  // k = 1: the reference's std() is NaN
    hipStream_t st = (hipStream_t)stream;
    const long rows = (long)B * N;
    const dim3 grid((unsigned)((rows + kThreads - 1) / kThreads));
    if (k <= 16)
        hipLaunchKernelGGL(structure_features_kernel<16>, grid, dim3(kThreads), 0, st, xyz, idx, B, N, k, feat, rel);
    else
        hipLaunchKernelGGL(structure_features_kernel<32>, grid, dim3(kThreads), 0, st, xyz, idx, B, N, k, feat, rel);
    return pcb_check_launch();
}
