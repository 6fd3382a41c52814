// fp32 row GEMMs of the shared pointwise MLPs for gfx950 -- the PARITY arithmetic of the engine:
// the same design as the bf16 kernels of gemm.hip (a layer keeps only y = x W^T; the BatchNorm /
// activation algebra of the neighbouring layers is applied on operand load, batch statistics and
// the BatchNorm-backward sums come out of the epilogues, weight gradients are split-row slabs
// summed in a fixed order), on fp32 rows and the fp32-input matrix cores
// (v_mfma_f32_32x32x2_f32: exact fp32, bit for bit an fma chain, 64 FLOP/clk/SIMD).
//
// Reference composition per layer (models/pointnet2_utils.py:149-151, :207-209, :353-354;
// models/DGCNN.py:19-30): Conv(1x1) -> BatchNorm -> ReLU/LeakyReLU in fp32.  With these kernels the
// fp32 mode of the package -- the one whose logits are held to 1e-4 of the reference's -- runs no
// library GEMM and no ATen BatchNorm.
//
//   gemm_nt  out[R,N] = A'[R,K] . W[N,K]^T    A' = A | act(A*scale+shift) | dy(dz,y) | dy(dout,argmax,y)
//   gemm_tn  dW[M,N]  = A'[R,M]^T . B'[R,N]   rows split over workgroups, per-split slabs
//
// A 16-byte operand chunk is 4 columns here (8 in bf16); a 32-float k-stage is 128 bytes per row,
// exactly the bf16 kernel's 64-element stage, so the staging pattern is the same.  One MFMA consumes
// two k values (lane half h supplies k = h); a lane reads 4 consecutive k of its row with one
// ds_read_b128 and feeds four MFMAs from it, so MFMA e of a group multiplies k = {e, 4 + e}: the
// sum over a stage is complete, only its order differs from left-to-right (the reference's sgemm
// order is unspecified anyway).
#include "gemm_shared.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { PRO_PLAIN = PCB_PRO_PLAIN, PRO_BNACT = PCB_PRO_BNACT, PRO_DY = PCB_PRO_DY, PRO_DY_POOL = PCB_PRO_DY_POOL };

__device__ __forceinline__ float act_slope(int act) { return act == 1 ? 0.0f : (act == 2 ? 0.2f : 1.0f); }
__device__ __forceinline__ float act_fwd(float u, float slope) { return u > 0.0f ? u : fmaf(slope, u, 0.0f); }
__device__ __forceinline__ float act_grad(float u, float slope) { return u > 0.0f ? 1.0f : slope; }

struct OperandF {
    const float *a0;          // PLAIN/BNACT: the rows; DY: dz rows; DY_POOL: unused
    const float *a1;          // DY / DY_POOL: y rows
    long ld;                  // row stride in elements (same for a0 and a1)
    const float *scale, *shift, *p, *q;   // per column
    const float *dout;        // DY_POOL: [groups, cols] fp32
    const unsigned char *arg; // DY_POOL: [groups, cols] uint8
    int ns;                   // DY_POOL: rows per group
    int act;
};

__device__ __forceinline__ OperandF local_copy(const OperandF &k)
{
    OperandF o;
    o.a0 = k.a0; o.a1 = k.a1; o.ld = k.ld;
    o.scale = k.scale; o.shift = k.shift; o.p = k.p; o.q = k.q;
    o.dout = k.dout; o.arg = k.arg; o.ns = k.ns; o.act = k.act;
    return o;
}

// Per-column constants of one 4-column chunk.
template <int PRO>
struct ConstsF {
    float4 scale, shift, p, q;
    __device__ __forceinline__ void load(const OperandF &o, int c, int cols)
    {
        if (PRO == PRO_PLAIN) return;
        const int cs = c < cols ? c : 0;  // chunks past the matrix are zeroed by RawF::finish
        scale = *reinterpret_cast<const float4 *>(o.scale + cs);
        shift = *reinterpret_cast<const float4 *>(o.shift + cs);
        if (PRO >= PRO_DY) {
            p = *reinterpret_cast<const float4 *>(o.p + cs);
            q = *reinterpret_cast<const float4 *>(o.q + cs);
        }
    }
};

// The untransformed bytes of one chunk (issued early), and finish() = the prologue math.
template <int PRO>
struct RawF {
    float4 v0;                // PLAIN/BNACT: rows; DY: dz
    float4 v1;                // DY / DY_POOL: y
    uint32_t arg;             // DY_POOL: 4 arg-max bytes
    const float *dptr;        // DY_POOL: this chunk's 4 dout values
    int j;                    // DY_POOL: row index inside its group
    bool live;
    __device__ __forceinline__ void load(const OperandF &o, long r, int c, long rows, int cols)
    {
        live = r < rows && c < cols;
        const long rs = r < rows ? r : rows - 1;
        const int cs = c < cols ? c : 0;
        if (PRO != PRO_DY_POOL) v0 = *reinterpret_cast<const float4 *>(o.a0 + rs * o.ld + cs);
        if (PRO >= PRO_DY) v1 = *reinterpret_cast<const float4 *>(o.a1 + rs * o.ld + cs);
        if (PRO == PRO_DY_POOL) {
            const long g = rs / o.ns;
            j = (int)(rs - g * o.ns);
            arg = *reinterpret_cast<const uint32_t *>(o.arg + g * cols + cs);
            dptr = o.dout + g * cols + cs;
        }
    }
    __device__ __forceinline__ float4 finish(const ConstsF<PRO> &k, float slope) const
    {
        if (!live) return make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (PRO == PRO_PLAIN) return v0;
        const float sc[4] = {k.scale.x, k.scale.y, k.scale.z, k.scale.w};
        const float sh[4] = {k.shift.x, k.shift.y, k.shift.z, k.shift.w};
        float f[4];
        if (PRO == PRO_BNACT) {
            const float a[4] = {v0.x, v0.y, v0.z, v0.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) f[i] = act_fwd(fmaf(a[i], sc[i], sh[i]), slope);
            return make_float4(f[0], f[1], f[2], f[3]);
        }
        const float y[4] = {v1.x, v1.y, v1.z, v1.w};
        if (PRO == PRO_DY) {
            f[0] = v0.x; f[1] = v0.y; f[2] = v0.z; f[3] = v0.w;
        } else {
            const float4 d4 = *reinterpret_cast<const float4 *>(dptr);
            const float d[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) f[i] = ((int)((arg >> (8 * i)) & 0xff) == j) ? d[i] : 0.0f;
        }
        const float pp[4] = {k.p.x, k.p.y, k.p.z, k.p.w};
        const float qq[4] = {k.q.x, k.q.y, k.q.z, k.q.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float du = f[i] * act_grad(fmaf(y[i], sc[i], sh[i]), slope);
            f[i] = fmaf(sc[i], du, fmaf(pp[i], y[i], qq[i]));
        }
        return make_float4(f[0], f[1], f[2], f[3]);
    }
};

// ---- gemm_nt ----------------------------------------------------------------------------------
constexpr int FN_BM = PCB_NT_BM, FN_BN = PCB_NT_BN, FN_BK = 32;
constexpr int FN_LD = FN_BK + 4;  // LDS row stride in floats (144 B): conflict-free ds_read_b128

// Optional epilogue of an input-gradient GEMM (see gemm.hip RedArgs): the tile just produced is dz
// of the layer below; (sum du, sum du*xhat) of ITS BatchNorm backward are accumulated here.
struct RedArgsF {
    const float *y;                              // [R, N]: pre-BatchNorm output of the layer below
    const float *scale, *shift, *mean, *invstd;  // its per-column constants
    int act;
    const float *bias;  // plain epilogue only (no STATS, no RED): out = A'.W^T + bias[N]
};

template <int PRO, int STATS, int RED>
__global__ __launch_bounds__(256, (RED ? 1 : 2)) void gemm_nt_f32_kernel(OperandF A_arg, const float *__restrict__ Bw, long R,
                                                              int N, int K, float *__restrict__ out,
                                                              float *__restrict__ sums, RedArgsF red)
{
    const OperandF A = local_copy(A_arg);
    const float a_slope = act_slope(A.act), red_slope = act_slope(red.act);
    __shared__ __attribute__((aligned(16))) float As[FN_BM * FN_LD];
    __shared__ __attribute__((aligned(16))) float Bs[FN_BN * FN_LD];
    __shared__ float ssum[4 * 2 * FN_BN];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int r31 = lane & 31, h = lane >> 5;
    const int n0 = blockIdx.y * FN_BN;
    const int chunk = t & 7;   // 4-column chunk inside a BK stage
    const int rrow = t >> 3;   // 0..31
    const long tiles_m = (R + FN_BM - 1) / FN_BM;

    long tile = blockIdx.x;
    if (tile >= tiles_m) {
        // more slabs than row tiles: this workgroup has no rows, its slab must still read as zero
        if ((STATS || RED) && t < FN_BN && n0 + t < N) {
            sums[((long)blockIdx.x * 2 + 0) * N + n0 + t] = 0.0f;
            sums[((long)blockIdx.x * 2 + 1) * N + n0 + t] = 0.0f;
        }
        return;
    }

    f32x16 acc[4];
    // each lane owns one output column per 32-wide tile for the whole kernel: column sums in registers
    float st_s[4] = {0.0f, 0.0f, 0.0f, 0.0f}, st_q[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    float rsc[4], rsh[4], rmu[4], ris[4];  // RED: constants of this lane's 4 columns
    if (RED) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + j * 32 + r31;
            const bool ok = n < N;
            rsc[j] = ok ? red.scale[n] : 0.0f;
            rsh[j] = ok ? red.shift[n] : 0.0f;
            rmu[j] = ok ? red.mean[n] : 0.0f;
            ris[j] = ok ? red.invstd[n] : 0.0f;
        }
    }

    RawF<PRO> ra[4];
    ConstsF<PRO> ka;
    float4 rb[4];
    bool liveb[4];
    auto fetch = [&](long tl, int kb) {
        const int kc = kb + chunk * 4;
        const long mb = tl * FN_BM;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ra[i].load(A, mb + rrow + 32 * i, kc, R, K);
            const int n = n0 + rrow + 32 * i;
            rb[i] = *reinterpret_cast<const float4 *>(Bw + (long)(n < N ? n : N - 1) * K + (kc < K ? kc : 0));
            liveb[i] = n < N && kc < K;
        }
    };
    fetch(tile, 0);
    for (; tile < tiles_m; tile += gridDim.x) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[j][i] = 0.0f;
        for (int k0 = 0; k0 < K; k0 += FN_BK) {
            ka.load(A, k0 + chunk * 4, K);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                *reinterpret_cast<float4 *>(&As[(rrow + 32 * i) * FN_LD + chunk * 4]) = ra[i].finish(ka, a_slope);
                *reinterpret_cast<float4 *>(&Bs[(rrow + 32 * i) * FN_LD + chunk * 4]) =
                    liveb[i] ? rb[i] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
            __syncthreads();
            // next stage's global loads fly under the MFMAs, across the tile boundary too
            const bool last_k = k0 + FN_BK >= K;
            const long ntile = last_k ? tile + gridDim.x : tile;
            const int nk = last_k ? 0 : k0 + FN_BK;
            if (ntile < tiles_m) fetch(ntile, nk);
#pragma unroll
            for (int g = 0; g < FN_BK / 8; ++g) {
                const int kk = g * 8 + 4 * h;
                const float4 fa4 = *reinterpret_cast<const float4 *>(&As[(wave * 32 + r31) * FN_LD + kk]);
                const float fa[4] = {fa4.x, fa4.y, fa4.z, fa4.w};
                float fb[4][4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 b4 = *reinterpret_cast<const float4 *>(&Bs[(j * 32 + r31) * FN_LD + kk]);
                    fb[j][0] = b4.x; fb[j][1] = b4.y; fb[j][2] = b4.z; fb[j][3] = b4.w;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[e], fb[j][e], acc[j], 0, 0, 0);
            }
            __syncthreads();
        }
        // tile done.  C/D map of a 32x32 tile: col = lane & 31, row = (i & 3) + 8*(i >> 2) + 4*(lane >> 5):
        // 32 lanes hold 32 consecutive fp32 columns of one row -- a full 128-byte line per store.
        const long m0 = tile * FN_BM + wave * 32;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + j * 32 + r31;
            const bool ncol = n < N;
            float bj = 0.0f;
            if (!STATS && !RED && red.bias && ncol) bj = red.bias[n];
            float sv = 0.0f, sq = 0.0f, r1 = 0.0f, r2 = 0.0f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const long r = m0 + (i & 3) + 8 * (i >> 2) + 4 * h;
                const float v = acc[j][i] + bj;
                if (r < R && ncol) out[r * N + n] = v;
                if (STATS) {  // rows past R and columns past N hold exact zeros: no mask needed
                    sv += v;
                    sq = fmaf(v, v, sq);
                }
                if (RED) {
                    const float yv = red.y[(r < R ? r : R - 1) * N + (ncol ? n : 0)];
                    const float du = v * act_grad(fmaf(yv, rsc[j], rsh[j]), red_slope);
                    r1 += du;
                    r2 = fmaf(du, (yv - rmu[j]) * ris[j], r2);
                }
            }
            if (STATS) {
                st_s[j] += sv;
                st_q[j] += sq;
            }
            if (RED) {
                st_s[j] += r1;
                st_q[j] += r2;
            }
        }
    }
    if (STATS || RED) {
        // lanes l and l+32 hold the same column (different rows); then the 4 waves meet in LDS; the
        // workgroup's totals go to ITS slab (no atomics: the finalize kernels sum the slabs in order)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float s2 = st_s[j] + __shfl_xor(st_s[j], 32);
            const float q2 = st_q[j] + __shfl_xor(st_q[j], 32);
            if (lane < 32) {
                ssum[(wave * 2 + 0) * FN_BN + j * 32 + lane] = s2;
                ssum[(wave * 2 + 1) * FN_BN + j * 32 + lane] = q2;
            }
        }
        __syncthreads();
        if (t < FN_BN && n0 + t < N) {
            float a = 0.0f, b = 0.0f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                a += ssum[(w * 2 + 0) * FN_BN + t];
                b += ssum[(w * 2 + 1) * FN_BN + t];
            }
            sums[((long)blockIdx.x * 2 + 0) * N + n0 + t] = a;
            sums[((long)blockIdx.x * 2 + 1) * N + n0 + t] = b;
        }
    }
}

// ---- gemm_tn (weight gradient) ----------------------------------------------------------------
// The fp32 MFMA takes its A operand as A[i = lane & 31][k = lane >> 5]: with the reduction index
// k = row and i = an output row of dW (a column of A'), 32 lanes read 32 consecutive floats of one
// staged row -- no transposed read needed (the bf16 kernel needs ds_read_tr16).
constexpr int FT_BM = PCB_TN_BM, FT_BN = PCB_TN_BN, FT_RS = PCB_TN_RS;
constexpr int FT_LD = 128 + 4;

__device__ __forceinline__ unsigned xcd_logical(unsigned id, unsigned total)
{
    const unsigned xcd = id & 7u, slot = id >> 3;
    const unsigned q = total >> 3, rem = total & 7u;
    return xcd * q + (xcd < rem ? xcd : rem) + slot;
}

template <int APRO, int BPRO>
__global__ __launch_bounds__(256, 2) void gemm_tn_f32_kernel(OperandF A_arg, OperandF B_arg, long R, int M, int N,
                                                              long rows_per_split, float *__restrict__ part,
                                                              int tiles_m, int tiles_n)
{
    const OperandF A = local_copy(A_arg);
    const OperandF B = local_copy(B_arg);
    const float a_slope = act_slope(A.act), b_slope = act_slope(B.act);
    __shared__ __attribute__((aligned(16))) float As[FT_RS * FT_LD];
    __shared__ __attribute__((aligned(16))) float Bs[FT_RS * FT_LD];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int r31 = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;  // 2 x 2 waves, 64 x 64 outputs each
    const unsigned logical = xcd_logical(blockIdx.x, gridDim.x);
    const int tile_n = logical % tiles_n;
    const int tile_m = (logical / tiles_n) % tiles_m;
    const int split = logical / (tiles_n * tiles_m);
    const int m0 = tile_m * FT_BM;
    const int n0 = tile_n * FT_BN;
    const long r_begin = (long)split * rows_per_split;
    const long r_end = r_begin + rows_per_split < R ? r_begin + rows_per_split : R;

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.0f;

    // staging: RS rows x 32 chunks of 4 columns per operand -> 4 chunks per thread and operand;
    // a thread's column chunk never changes, so its per-column constants stay in registers
    const int chunk = t & 31;
    const int rrow = t >> 5;  // 0..7
    ConstsF<APRO> ka;
    ConstsF<BPRO> kb;
    ka.load(A, m0 + chunk * 4, M);
    kb.load(B, n0 + chunk * 4, N);
    RawF<APRO> ra[4];
    RawF<BPRO> rb[4];
    auto fetch = [&](long r0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long r = r0 + rrow + 8 * i;
            ra[i].load(A, r < r_end ? r : R, m0 + chunk * 4, R, M);
            rb[i].load(B, r < r_end ? r : R, n0 + chunk * 4, R, N);
        }
    };
    if (r_begin < r_end) fetch(r_begin);
    for (long r0 = r_begin; r0 < r_end; r0 += FT_RS) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<float4 *>(&As[(rrow + 8 * i) * FT_LD + chunk * 4]) = ra[i].finish(ka, a_slope);
            *reinterpret_cast<float4 *>(&Bs[(rrow + 8 * i) * FT_LD + chunk * 4]) = rb[i].finish(kb, b_slope);
        }
        __syncthreads();
        if (r0 + FT_RS < r_end) fetch(r0 + FT_RS);
#pragma unroll
        for (int ks = 0; ks < FT_RS / 2; ++ks) {
            float af[2], bf[2];
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                af[x] = As[(2 * ks + h) * FT_LD + wm * 64 + x * 32 + r31];
                bf[x] = Bs[(2 * ks + h) * FT_LD + wn * 64 + x * 32 + r31];
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
        __syncthreads();
    }

#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int n = n0 + wn * 64 + b * 32 + r31;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int m = m0 + wm * 64 + a * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                if (m < M && n < N) part[((long)split * M + m) * N + n] = acc[a][b][i];
            }
        }
}

template <int PRO>
void launch_nt(const OperandF &A, const float *Bw, long R, int N, int K, float *out, float *sums, int nparts,
               hipStream_t st, const RedArgsF &red, bool with_red)
{
    const unsigned ny = (unsigned)((N + FN_BN - 1) / FN_BN);
    const dim3 grid((unsigned)(sums ? nparts : pcb_nt_grid_x(PRO, R, N, pcb_busy_cus())), ny);
    if (with_red && PRO >= PRO_DY)
        hipLaunchKernelGGL((gemm_nt_f32_kernel<PRO, 0, (PRO >= PRO_DY)>), grid, dim3(256), 0, st, A, Bw, R, N, K, out, sums, red);
    else if (sums)
        hipLaunchKernelGGL((gemm_nt_f32_kernel<PRO, 1, 0>), grid, dim3(256), 0, st, A, Bw, R, N, K, out, sums, red);
    else
        hipLaunchKernelGGL((gemm_nt_f32_kernel<PRO, 0, 0>), grid, dim3(256), 0, st, A, Bw, R, N, K, out, sums, red);
}

template <int APRO>
void launch_tn(const OperandF &A, const OperandF &B, int bpro, long R, int M, int N, float *part, float *dW,
               int out_cols, int out_perm, hipStream_t st)
{
    long rps;
    const long splits = pcb_tn_splits(R, M, N, &rps, 512 - 2 * pcb_busy_cus());
    const int tm = (M + FT_BM - 1) / FT_BM, tn = (N + FT_BN - 1) / FT_BN;
    const dim3 grid((unsigned)(tm * tn * splits));
    if (bpro == PRO_PLAIN)
        hipLaunchKernelGGL((gemm_tn_f32_kernel<APRO, PRO_PLAIN>), grid, dim3(256), 0, st, A, B, R, M, N, rps, part, tm, tn);
    else
        hipLaunchKernelGGL((gemm_tn_f32_kernel<APRO, PRO_BNACT>), grid, dim3(256), 0, st, A, B, R, M, N, rps, part, tm, tn);
    (void)pcb_reduce_slabs(part, (int)splits, (long)M * N, dW, N, out_cols, out_perm, 4, st);
}

inline bool bad_dim(long v) { return v <= 0 || (v & 3) != 0; }

OperandF make_operand(const void *a0, const void *a1, long ld, const float *scale, const float *shift, const float *p,
                      const float *q, const float *dout, const unsigned char *arg, int ns, int act)
{
    OperandF o;
    o.a0 = (const float *)a0;
    o.a1 = (const float *)a1;
    o.ld = ld;
    o.scale = scale;
    o.shift = shift;
    o.p = p;
    o.q = q;
    o.dout = dout;
    o.arg = arg;
    o.ns = ns > 0 ? ns : 1;
    o.act = act;
    return o;
}

// Algorithmic HBM bytes of one gemm_nt launch (fp32 rows): the A operand as its prologue reads it + the
// output (+ the y rows of the RED epilogue); weights and per-channel vectors not counted.
inline double nt_bytes(int pro, long R, int N, int K, int ns, bool with_red)
{
    double a = 4.0 * R * K;
    if (pro == PRO_DY) a = 8.0 * R * K;
    if (pro == PRO_DY_POOL) a = 4.0 * R * K + 5.0 * (double)(R / (ns > 0 ? ns : 1)) * K;
    return a + 4.0 * R * N * (with_red ? 2.0 : 1.0);
}

int nt_dispatch(int pro, const OperandF &A, const float *w, long R, int N, int K, float *out, float *sums, int nparts,
                hipStream_t st, const RedArgsF &red, bool with_red)
{
    hipEvent_t timed;
    pcb_timer_begin(st, &timed);
    switch (pro) {
        case PRO_PLAIN: launch_nt<PRO_PLAIN>(A, w, R, N, K, out, sums, nparts, st, red, with_red); break;
        case PRO_BNACT: launch_nt<PRO_BNACT>(A, w, R, N, K, out, sums, nparts, st, red, with_red); break;
        case PRO_DY: launch_nt<PRO_DY>(A, w, R, N, K, out, sums, nparts, st, red, with_red); break;
        default: launch_nt<PRO_DY_POOL>(A, w, R, N, K, out, sums, nparts, st, red, with_red); break;
    }
    pcb_timer_end(st, timed, nt_bytes(pro, R, N, K, A.ns, with_red), pro + (with_red ? 10 : 0), R, N, K);
    return pcb_check_launch();
}

}  // namespace

extern "C" int pcb_gemm_nt_f32(int pro, const void *a0, const void *a1, const float *scale, const float *shift,
                               const float *p, const float *q, const float *dout, const unsigned char *argmax, int ns,
                               int act, const void *w, long R, int N, int K, void *out, float *sums, int nparts,
                               void *stream)
{
    if (!w || !out || R <= 0) return PCB_ERR_INVALID_ARG;
    if (sums && (nparts < 1 || nparts > PCB_MAX_SLABS)) return PCB_ERR_INVALID_ARG;
    if (bad_dim(N) || bad_dim(K)) return PCB_ERR_UNSUPPORTED;
    if (pro < 0 || pro > 3) return PCB_ERR_INVALID_ARG;
    if ((pro <= PRO_DY && !a0) || (pro >= PRO_DY && !a1) || (pro >= PRO_BNACT && (!scale || !shift))) return PCB_ERR_INVALID_ARG;
    if (pro >= PRO_DY && (!p || !q)) return PCB_ERR_INVALID_ARG;
    if (pro == PRO_DY_POOL && (!dout || !argmax || ns <= 0 || ns > 255)) return PCB_ERR_INVALID_ARG;
    const OperandF A = make_operand(a0, a1, K, scale, shift, p, q, dout, argmax, ns, act);
    const RedArgsF none = {nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr};
    return nt_dispatch(pro, A, (const float *)w, R, N, K, (float *)out, sums, nparts, (hipStream_t)stream, none, false);
}

// y = x W^T + b for a conv without BatchNorm (fp32 rows).
extern "C" int pcb_gemm_nt_bias_f32(const void *a, const void *w, const float *bias, long R, int N, int K, void *out,
                                    void *stream)
{
    if (!a || !w || !out || R <= 0) return PCB_ERR_INVALID_ARG;
    if (bad_dim(N) || bad_dim(K)) return PCB_ERR_UNSUPPORTED;
    const OperandF A = make_operand(a, nullptr, K, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1, 0);
    const RedArgsF epi = {nullptr, nullptr, nullptr, nullptr, nullptr, 0, bias};
    return nt_dispatch(PRO_PLAIN, A, (const float *)w, R, N, K, (float *)out, nullptr, 0, (hipStream_t)stream, epi, false);
}

extern "C" int pcb_gemm_nt_red_f32(int pro, const void *a0, const void *a1, const float *scale, const float *shift,
                                   const float *p, const float *q, const float *dout, const unsigned char *argmax,
                                   int ns, int act, const void *w, long R, int N, int K, void *out, const void *red_y,
                                   const float *red_scale, const float *red_shift, const float *red_mean,
                                   const float *red_invstd, int red_act, float *red_sums, int nparts, void *stream)
{
    if (!w || !out || R <= 0 || !red_y || !red_scale || !red_shift || !red_mean || !red_invstd || !red_sums)
        return PCB_ERR_INVALID_ARG;
    if (nparts < 1 || nparts > PCB_MAX_SLABS) return PCB_ERR_INVALID_ARG;
    if (bad_dim(N) || bad_dim(K)) return PCB_ERR_UNSUPPORTED;
    if (pro != PRO_DY && pro != PRO_DY_POOL) return PCB_ERR_INVALID_ARG;
    if (!a1 || !scale || !shift || !p || !q || (pro == PRO_DY && !a0)) return PCB_ERR_INVALID_ARG;
    if (pro == PRO_DY_POOL && (!dout || !argmax || ns <= 0 || ns > 255)) return PCB_ERR_INVALID_ARG;
    const OperandF A = make_operand(a0, a1, K, scale, shift, p, q, dout, argmax, ns, act);
    const RedArgsF red = {(const float *)red_y, red_scale, red_shift, red_mean, red_invstd, red_act, nullptr};
    return nt_dispatch(pro, A, (const float *)w, R, N, K, (float *)out, red_sums, nparts, (hipStream_t)stream, red, true);
}

extern "C" int pcb_gemm_tn_f32(int apro, const void *dz, const void *y, const float *scale, const float *shift,
                               const float *p, const float *q, const float *dout, const unsigned char *argmax, int ns,
                               int act, int bpro, const void *x, const float *xscale, const float *xshift, int xact,
                               long R, int M, int N, float *workspace, float *dW, int out_cols, int out_perm,
                               void *stream)
{
    if (!x || !dW || !workspace || R <= 0) return PCB_ERR_INVALID_ARG;
    if (out_cols <= 0) { out_cols = N; out_perm = 0; }
    if (out_cols > N) return PCB_ERR_INVALID_ARG;
    if (bad_dim(M) || bad_dim(N)) return PCB_ERR_UNSUPPORTED;
    if (apro != PRO_PLAIN && apro != PRO_DY && apro != PRO_DY_POOL) return PCB_ERR_INVALID_ARG;
    if (apro != PRO_PLAIN && (!y || !scale || !shift || !p || !q)) return PCB_ERR_INVALID_ARG;
    if (apro == PRO_DY_POOL ? (!dout || !argmax || ns <= 0 || ns > 255) : !dz) return PCB_ERR_INVALID_ARG;
    if (bpro != PRO_PLAIN && bpro != PRO_BNACT) return PCB_ERR_INVALID_ARG;
    if (bpro == PRO_BNACT && (!xscale || !xshift)) return PCB_ERR_INVALID_ARG;
    const OperandF A = make_operand(dz, y, M, scale, shift, p, q, dout, argmax, ns, act);
    const OperandF B = make_operand(x, nullptr, N, xscale, xshift, nullptr, nullptr, nullptr, nullptr, 1, xact);
    hipStream_t st = (hipStream_t)stream;
    if (apro == PRO_PLAIN)
        launch_tn<PRO_PLAIN>(A, B, bpro, R, M, N, workspace, dW, out_cols, out_perm, st);
    else if (apro == PRO_DY)
        launch_tn<PRO_DY>(A, B, bpro, R, M, N, workspace, dW, out_cols, out_perm, st);
    else
        launch_tn<PRO_DY_POOL>(A, B, bpro, R, M, N, workspace, dW, out_cols, out_perm, st);
    pcb_account((apro == PRO_DY ? 8.0 : 4.0) * R * M + (apro == PRO_DY_POOL ? 5.0 * (double)(R / (ns > 0 ? ns : 1)) * M : 0.0) + 4.0 * R * N);
    return pcb_check_launch();
}
