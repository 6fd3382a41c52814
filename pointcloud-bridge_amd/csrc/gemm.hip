// bf16 MFMA row GEMMs of the shared pointwise MLPs for gfx950, with the BatchNorm / activation
// algebra of the neighbouring layers fused into operand loads and epilogues.
//
// Reference composition per layer (models/pointnet2_utils.py:149-151, :207-209, :353-354;
// models/DGCNN.py:19-30): Conv(1x1) -> BatchNorm -> ReLU/LeakyReLU on fp32 [B,C,S,ns] tensors,
// i.e. one GEMM plus 5-7 elementwise/reduction passes, and the same again in backward.
// Here a layer keeps only y = x W^T (bf16 rows).  Everything else is recomputed on the fly:
//
//   gemm_nt  out[R,N] = A'[R,K] . B[N,K]^T          (forward and input-gradient GEMM)
//       A' = A                                   plain rows (grouped input)
//          | act(A*scale + shift)                previous layer's BatchNorm+activation on load
//          | dy(dz, y)                           BatchNorm/activation BACKWARD on load (dense dz)
//          | dy(dout, argmax, y)                 the same for a max-pooled layer
//       epilogue: bf16 store; optionally per-column sum / sum of squares (next BatchNorm's
//       batch statistics) accumulated from the rounded outputs
//   gemm_tn  dW[M,N]  = A'[R,M]^T . B'[R,N]        (weight gradient; rows split over workgroups, per-split
//                                                  slabs summed in a fixed order -- no atomics)
//       A' = dy(...) as above, B' = plain rows | act(B*scale + shift)
//
// dy = scale*du + p*y + q with du = dz*act'(y*scale+shift); p, q fold the batch-statistics terms
// (-s1/R - xhat*s2/R) and come from pcb_bn_bwd_finalize.
//
// These GEMMs are skinny (K, N <= a few hundred, R up to ~10^6): HBM-bound, so the design goal is
// one coalesced pass over each activation with full 128-byte rows staged through LDS, fp32
// accumulation in v_mfma_f32_32x32x16_bf16, and no intermediate tensor.  gemm_tn needs both
// operands transposed (reduction index = row = slow memory axis): tiles are stored row-major as
// loaded and read back with ds_read_b64_tr_b16 (hardware transpose read).
#include <stdio.h>
#include <stdlib.h>

#include "gemm_shared.h"

namespace {

typedef unsigned short u16;
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u16 f2bf(float f) { return __builtin_bit_cast(u16, (__bf16)f); }
__device__ __forceinline__ float bf2f(u16 h) { return __uint_as_float((uint32_t)h << 16); }
__device__ __forceinline__ void unpack8(const uint4 &v, float *f)
{
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
    f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 pack8(const float *f)
{
    uint4 v;
    v.x = (uint32_t)f2bf(f[0]) | ((uint32_t)f2bf(f[1]) << 16);
    v.y = (uint32_t)f2bf(f[2]) | ((uint32_t)f2bf(f[3]) << 16);
    v.z = (uint32_t)f2bf(f[4]) | ((uint32_t)f2bf(f[5]) << 16);
    v.w = (uint32_t)f2bf(f[6]) | ((uint32_t)f2bf(f[7]) << 16);
    return v;
}
// Activations as one select on a per-launch slope (0 ReLU, 0.2 LeakyReLU, 1 none): testing the
// activation code per element would put scalar branches into the innermost prologue loops.
__device__ __forceinline__ float act_slope(int act) { return act == 1 ? 0.0f : (act == 2 ? 0.2f : 1.0f); }
__device__ __forceinline__ float act_grad(float u, float slope) { return u > 0.0f ? 1.0f : slope; }

// ---- operand prologues ------------------------------------------------------------------------
enum { PRO_PLAIN = 0, PRO_BNACT = 1, PRO_DY = 2, PRO_DY_POOL = 3 };

struct Operand {
    const u16 *a0;            // PLAIN/BNACT: the rows; DY: dz rows; DY_POOL: unused
    const u16 *a1;            // DY / DY_POOL: y rows
    long ld;                  // row stride in elements (same for a0 and a1)
    const float *scale, *shift, *p, *q;   // per column
    const float *dout;        // DY_POOL: [groups, cols] fp32
    const unsigned char *arg; // DY_POOL: [groups, cols] uint8
    int ns;                   // DY_POOL: rows per group
    int act;
};

// Kernel-argument structs live in the kernarg segment; taking a reference to one makes the compiler
// copy it to scratch.  A field-by-field copy into a local lets it dissolve into SGPRs instead.
__device__ __forceinline__ Operand local_copy(const Operand &k)
{
    Operand o;
    o.a0 = k.a0; o.a1 = k.a1; o.ld = k.ld;
    o.scale = k.scale; o.shift = k.shift; o.p = k.p; o.q = k.q;
    o.dout = k.dout; o.arg = k.arg; o.ns = k.ns; o.act = k.act;
    return o;
}

// An operand chunk = 8 consecutive columns [c, c+8) of one row.  Loading is split in two so that a
// stage's global loads can be in flight while the previous stage's MFMAs run:
//   Raw<PRO>     the untransformed bytes of a chunk (issued early, no dependent math)
//   Consts<PRO>  the per-column fp32 constants of the chunk's 8 columns
//   finish()     the prologue math, producing the 8 packed bf16 values that go to LDS
template <int PRO>
struct Consts {
    float scale[8], shift[8], p[8], q[8];
    __device__ __forceinline__ void load(const Operand &o, int c, int cols)
    {
        if (PRO == PRO_PLAIN) return;
        // unconditional 16-byte loads from a clamped chunk (cols % 8 == 0, so chunk 0 always exists);
        // chunks past the matrix are zeroed by Raw::finish, whatever constants they meet
        const int cs = c < cols ? c : 0;
        load8(o.scale + cs, scale);
        load8(o.shift + cs, shift);
        if (PRO >= PRO_DY) {
            load8(o.p + cs, p);
            load8(o.q + cs, q);
        }
    }
    static __device__ __forceinline__ void load8(const float *src, float *dst)
    {
        const float4 a = *reinterpret_cast<const float4 *>(src);
        const float4 b = *reinterpret_cast<const float4 *>(src + 4);
        dst[0] = a.x; dst[1] = a.y; dst[2] = a.z; dst[3] = a.w;
        dst[4] = b.x; dst[5] = b.y; dst[6] = b.z; dst[7] = b.w;
    }
};

template <int PRO>
struct Raw {
    uint4 v0;                 // PLAIN/BNACT: rows; DY: dz
    uint4 v1;                 // DY / DY_POOL: y
    unsigned long long arg;   // DY_POOL: packed arg-max bytes
    const float *dptr;        // DY_POOL: this chunk's 8 dout values (read in finish(): the group's
                              // dout row is shared by ns consecutive rows, so it sits in L1/L2)
    int j;                    // DY_POOL: row index inside its group
    bool live;
    __device__ __forceinline__ void load(const Operand &o, long r, int c, long rows, int cols)
    {
        // Loads are unconditional (from a clamped, always valid address) so that they stay plain
        // register loads; out-of-range chunks are zeroed in finish().
        live = r < rows && c < cols;
        const long rs = r < rows ? r : rows - 1;
        const int cs = c < cols ? c : 0;
        if (PRO != PRO_DY_POOL) v0 = *reinterpret_cast<const uint4 *>(o.a0 + rs * o.ld + cs);
        if (PRO >= PRO_DY) v1 = *reinterpret_cast<const uint4 *>(o.a1 + rs * o.ld + cs);
        if (PRO == PRO_DY_POOL) {
            const long g = rs / o.ns;
            j = (int)(rs - g * o.ns);
            arg = *reinterpret_cast<const unsigned long long *>(o.arg + g * cols + cs);
            dptr = o.dout + g * cols + cs;
        }
    }
    __device__ __forceinline__ uint4 finish(const Consts<PRO> &k, float slope) const
    {
        const uint32_t keep = live ? 0xffffffffu : 0u;
        if (PRO == PRO_PLAIN) return make_uint4(v0.x & keep, v0.y & keep, v0.z & keep, v0.w & keep);
        float f[8];
        if (PRO == PRO_BNACT) {
            // (packed pairs per dword, see PRO_DY below)
            const uint32_t xw[4] = {v0.x, v0.y, v0.z, v0.w};
            uint32_t o[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const f32x2 xv = {__uint_as_float(xw[d] << 16), __uint_as_float(xw[d] & 0xffff0000u)};
                const f32x2 sc = {k.scale[2 * d], k.scale[2 * d + 1]}, sh = {k.shift[2 * d], k.shift[2 * d + 1]};
                const f32x2 u = __builtin_elementwise_fma(xv, sc, sh);
                const f32x2 zero = {0.0f, 0.0f}, sl = {slope, slope};
                const f32x2 neg = __builtin_elementwise_fma(sl, u, zero);   // act_fwd: fmaf(slope, u, 0)
                const f32x2 r = {u.x > 0.0f ? u.x : neg.x, u.y > 0.0f ? u.y : neg.y};
                o[d] = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, bf16x2_t)) & keep;
            }
            return make_uint4(o[0], o[1], o[2], o[3]);
        }
        if (PRO == PRO_DY) {
            // the two values of one dword stay one packed-fp32 pair from unpack to pack: v_pk_fma / v_pk_mul on the
            // pair, one v_cvt_pk_bf16_f32 back into the dword (left to itself the vectoriser paired the low halves
            // of two dwords and re-shuffled the results: four more vector instructions per dword pair)
            const uint32_t zw[4] = {v0.x, v0.y, v0.z, v0.w}, yw[4] = {v1.x, v1.y, v1.z, v1.w};
            uint32_t o[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const f32x2 yv = {__uint_as_float(yw[d] << 16), __uint_as_float(yw[d] & 0xffff0000u)};
                const f32x2 zv = {__uint_as_float(zw[d] << 16), __uint_as_float(zw[d] & 0xffff0000u)};
                const f32x2 sc = {k.scale[2 * d], k.scale[2 * d + 1]}, sh = {k.shift[2 * d], k.shift[2 * d + 1]};
                const f32x2 pp = {k.p[2 * d], k.p[2 * d + 1]}, qq = {k.q[2 * d], k.q[2 * d + 1]};
                const f32x2 u = __builtin_elementwise_fma(yv, sc, sh);
                const f32x2 g = {u.x > 0.0f ? 1.0f : slope, u.y > 0.0f ? 1.0f : slope};
                const f32x2 du = zv * g;
                const f32x2 r = __builtin_elementwise_fma(sc, du, __builtin_elementwise_fma(pp, yv, qq));
                o[d] = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, bf16x2_t)) & keep;  // one v_cvt_pk_bf16_f32
            }
            return make_uint4(o[0], o[1], o[2], o[3]);
        }
        float y[8];
        unpack8(v1, y);
        if (PRO == PRO_DY) {
            unpack8(v0, f);
        } else {
            const float4 d0 = *reinterpret_cast<const float4 *>(dptr);
            const float4 d1 = *reinterpret_cast<const float4 *>(dptr + 4);
            const float d[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
#pragma unroll
            for (int i = 0; i < 8; ++i) f[i] = ((int)((arg >> (8 * i)) & 0xff) == j) ? d[i] : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float du = f[i] * act_grad(fmaf(y[i], k.scale[i], k.shift[i]), slope);
            f[i] = fmaf(k.scale[i], du, fmaf(k.p[i], y[i], k.q[i]));
        }
        const uint4 r = pack8(f);
        return make_uint4(r.x & keep, r.y & keep, r.z & keep, r.w & keep);
    }
};

// ---- gemm_nt ----------------------------------------------------------------------------------
constexpr int NT_BM = 128, NT_BN = 128, NT_BK = 64;
constexpr int NT_LD = NT_BK + 8;  // LDS row stride in bf16 (144 B): conflict-free ds_read_b128
constexpr int NT_OUT_LD = NT_BN + 8;  // row stride of the per-wave output staging block (272 B)
static_assert(4 * 32 * NT_OUT_LD <= (NT_BM + NT_BN) * NT_LD, "output staging must fit the stage buffers");

// Optional epilogue of an input-gradient GEMM: the tile it has just produced is dz of the layer
// below; with that layer's y and BatchNorm constants the sums s1 = sum du, s2 = sum du*xhat of ITS
// BatchNorm backward are accumulated here, which saves the separate reduction pass over (dz, y).
struct RedArgs {
    const u16 *y;                              // [R, N] bf16: pre-BatchNorm output of the layer below
    const float *scale, *shift, *mean, *invstd;  // its per-column constants
    int act;
    const float *bias;  // plain epilogue only (no STATS, no RED): out = A'.W^T + bias[N] (a conv without BatchNorm)
    const u16 *res;     // ... + res[R, N] (bf16 rows, added to the rounded result: the sum of two branches' outputs)
    // STATS == 2 (eight-wave form): out = A'.W^T + add1[r >> sh1, :] + add2[r >> sh2, :] before rounding -- fp32 rows
    // [R >> sh, N], each standing for 2^sh consecutive output rows (sh >= 2; add2 optional).  The layer's input is the
    // concatenation of a full-resolution level with coarser levels REPEATED along the rows
    // (MultiScaleFeatureFusion, models/model.py:150-170): the conv is linear, so the coarse levels' share of the
    // product is computed on their own rows and arrives here instead of as 2^sh copies in the A operand.
    const float *add1, *add2;
    int sh1, sh2;
    // STATS: out = bf16(A'.W^T [+ addends] - centre[N]) -- the layer's pre-BatchNorm rows are stored CENTRED on a per-column
    // constant close to their batch mean (BatchNorm is invariant to it: the finalize kernel works on the centred moments and
    // adds it back for running_mean).  A bf16 value carries an absolute error of 2^-9 |y| and BatchNorm divides by std(y): stored
    // uncentred, the error in units of the normalised signal is 2^-9 (|mean|/std + 1).  NULL: no centring.
    const float *centre;
};

template <int PRO, int STATS, int RED, int OUT32 = 0>
__global__ __launch_bounds__(256, (PRO <= PRO_BNACT ? 3 : 2)) void gemm_nt_kernel(Operand A_arg, const u16 *__restrict__ Bw, long R,
                                                       int N, int K, u16 *__restrict__ out,
                                                       float *__restrict__ sums, RedArgs red_arg)
{
    const Operand A = local_copy(A_arg);
    const float a_slope = act_slope(A.act), red_slope = act_slope(red_arg.act);
    RedArgs red;
    red.y = red_arg.y; red.scale = red_arg.scale; red.shift = red_arg.shift;
    red.mean = red_arg.mean; red.invstd = red_arg.invstd; red.act = red_arg.act;
    // one LDS array: [A stage | B stage] in the main loop, per-wave output staging in the epilogue
    __shared__ __attribute__((aligned(16))) u16 smem[(NT_BM + NT_BN) * NT_LD];
    u16 *const As = smem;
    u16 *const Bs = smem + NT_BM * NT_LD;
    __shared__ float ssum[4 * 2 * NT_BN];
    __shared__ __attribute__((aligned(16))) float rconst[RED ? 4 * NT_BN : 4];  // RED: scale|shift|mean|invstd of this column tile

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int n0 = blockIdx.y * NT_BN;
    const int chunk = t & 7;   // 8-column chunk inside a BK stage
    const int rrow = t >> 3;   // 0..31
    const long tiles_m = (R + NT_BM - 1) / NT_BM;

    // Persistent over row tiles: K is short (a few stages), so a workgroup that handled one tile
    // would spend its life in load latency.  The (tile, k-stage) pairs of all its tiles form one
    // stream of stages and the next stage's loads are always in flight under the current MFMAs,
    // across tile boundaries too.
    long tile = blockIdx.x;
    if (tile >= tiles_m) {
        // more slabs than row tiles: this workgroup has no rows, its slab must still read as zero
        if ((STATS || RED) && t < NT_BN && n0 + t < N) {
            sums[((long)blockIdx.x * 2 + 0) * N + n0 + t] = 0.0f;
            sums[((long)blockIdx.x * 2 + 1) * N + n0 + t] = 0.0f;
        }
        return;
    }

    f32x16 acc[4];
    // statistics: each lane owns one output column per 32-wide tile for the whole kernel, so the
    // column sums live in 8 registers and meet the other lanes/waves only once, at the very end
    float st_s[4] = {0.0f, 0.0f, 0.0f, 0.0f}, st_q[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (RED) {
        for (int i = t; i < 4 * 2 * NT_BN; i += 256) ssum[i] = 0.0f;  // per-wave (s1, s2) slabs
        if (t < NT_BN) {
            const bool ok = n0 + t < N;
            rconst[0 * NT_BN + t] = ok ? red.scale[n0 + t] : 0.0f;
            rconst[1 * NT_BN + t] = ok ? red.shift[n0 + t] : 0.0f;
            rconst[2 * NT_BN + t] = ok ? red.mean[n0 + t] : 0.0f;
            rconst[3 * NT_BN + t] = ok ? red.invstd[n0 + t] : 0.0f;
        }
    }

    Raw<PRO> ra[4];
    Consts<PRO> ka;
    uint4 rb[4];
    uint32_t keepb[4];
    auto fetch = [&](long tl, int kb) {
        const int kc = kb + chunk * 8;
        const long mb = tl * NT_BM;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ra[i].load(A, mb + rrow + 32 * i, kc, R, K);
            const int n = n0 + rrow + 32 * i;
            // raw load now, masking when the stage is written to LDS: touching the value here would
            // make the wave wait for the load before it reaches the MFMAs
            rb[i] = *reinterpret_cast<const uint4 *>(Bw + (long)(n < N ? n : N - 1) * K + (kc < K ? kc : 0));
            keepb[i] = (n < N && kc < K) ? 0xffffffffu : 0u;
        }
    };
    fetch(tile, 0);
    for (; tile < tiles_m; tile += gridDim.x) {
        // zeroed here, not in the epilogue: the accumulators are then dead during the read-back,
        // which leaves their registers to the RED epilogue's operands
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[j][i] = 0.0f;
        for (int k0 = 0; k0 < K; k0 += NT_BK) {
            // per-column constants of this stage's chunk: L1/L2-resident, fetched here rather than
            // with the prefetch so that they are not live across the MFMA section
            ka.load(A, k0 + chunk * 8, K);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                *reinterpret_cast<uint4 *>(&As[(rrow + 32 * i) * NT_LD + chunk * 8]) = ra[i].finish(ka, a_slope);
                *reinterpret_cast<uint4 *>(&Bs[(rrow + 32 * i) * NT_LD + chunk * 8]) =
                    make_uint4(rb[i].x & keepb[i], rb[i].y & keepb[i], rb[i].z & keepb[i], rb[i].w & keepb[i]);
            }
            __syncthreads();
            // next stage's global loads fly under the MFMAs, across the tile boundary too
            // (ONE call site: two would double the live staging registers)
            const bool last_k = k0 + NT_BK >= K;
            const long ntile = last_k ? tile + gridDim.x : tile;
            const int nk = last_k ? 0 : k0 + NT_BK;
            if (ntile < tiles_m) fetch(ntile, nk);
            // operand fragments of k-step ks+1 are read from LDS while the MFMAs of k-step ks run
            bf16x8 fa[2], fb[2][4];
            auto frags = [&](int buf, int ks) {
                const int kk = ks * 16 + (lane >> 5) * 8;
                fa[buf] = *reinterpret_cast<const bf16x8 *>(&As[(wave * 32 + (lane & 31)) * NT_LD + kk]);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    fb[buf][j] = *reinterpret_cast<const bf16x8 *>(&Bs[(j * 32 + (lane & 31)) * NT_LD + kk]);
            };
            frags(0, 0);
#pragma unroll
            for (int ks = 0; ks < NT_BK / 16; ++ks) {
                if (ks + 1 < NT_BK / 16) frags((ks + 1) & 1, ks + 1);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks & 1], fb[ks & 1][j], acc[j], 0, 0, 0);
            }
            __syncthreads();
        }
        // tile done.  C/D map of a 32x32 tile: col = lane & 31, row = (i & 3) + 8*(i >> 2) + 4*(lane >> 5).
        // A lane owns single elements of many rows, so the tile goes through LDS once more: each
        // wave parks its 32 x 128 bf16 block in its own slice of the (now idle) stage buffers
        // and stores it back as 16-byte row segments -- 8 wide stores per lane instead of 64
        // two-byte ones.
        const long m0 = tile * NT_BM;
        if (OUT32) {
            // fp32 output (`out` then points to floats): straight from the accumulators, 32 lanes
            // writing 128 contiguous bytes of a row.  For the small per-point GEMMs whose results
            // are differenced afterwards (csrc/gatherlin.hip) and must not be rounded to bf16.
            float *const o32 = reinterpret_cast<float *>(out);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + j * 32 + (lane & 31);
                const float bj = (red_arg.bias && n < N) ? red_arg.bias[n] : 0.0f;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const long r = m0 + wave * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                    if (r < R && n < N) o32[r * N + n] = acc[j][i] + bj;
                }
            }
            __syncthreads();
            continue;
        }
        u16 *const stage = smem + wave * (32 * NT_OUT_LD);
        const long rows_here = R - m0 < NT_BM ? R - m0 : NT_BM;
        const __amdgpu_buffer_rsrc_t orsrc =
            __builtin_amdgcn_make_buffer_rsrc(out + m0 * N, 0, (int)(rows_here * N * 2), 0x00020000);
        // RED: the y rows of this tile are requested BEFORE the tile is stored.  (A load issued after
        // a store has to wait for the store's acknowledgement too -- the memory counter is shared --
        // and eight such round trips per tile cost more than the GEMM.)  Unconditional, from clamped
        // addresses: rows/columns outside the matrix meet dz = 0 below and contribute nothing.
        uint4 yraw[RED ? 8 : 1];
        if (RED) {
#pragma unroll
            for (int v = 0; v < 8; ++v) {
                const int ch = v * 64 + lane;
                const long r = m0 + wave * 32 + (ch >> 4);
                const int n = n0 + (ch & 15) * 8;
                yraw[v] = *reinterpret_cast<const uint4 *>(red.y + (r < R ? r : R - 1) * N + (n < N ? n : 0));
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float sv = 0.0f, sq = 0.0f;
            float bj = 0.0f;
            if (!STATS && !RED && red_arg.bias) {
                const int n = n0 + j * 32 + (lane & 31);
                bj = n < N ? red_arg.bias[n] : 0.0f;
            }
            if (STATS && red_arg.centre) {   // (wave-uniform) rows stored centred: see RedArgs
                const int n = n0 + j * 32 + (lane & 31);
                bj = n < N ? -red_arg.centre[n] : 0.0f;
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int rr = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                const u16 h = f2bf(!RED ? acc[j][i] + bj : acc[j][i]);
                stage[rr * NT_OUT_LD + j * 32 + (lane & 31)] = h;
                if (STATS) {
                    const float v = bf2f(h);  // statistics of the values the next kernels will read
                    sv += v;
                    sq = fmaf(v, v, sq);
                }
            }
            if (STATS) {
                st_s[j] += sv;
                st_q[j] += sq;
            }
        }
        // same wave wrote and reads its slice: a wave-level LDS fence is enough
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
        // every lane handles the same 8 columns in all of its 8 chunks: cc = (lane & 15) * 8
        float rs1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, rs2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int v = 0; v < 8; ++v) {
            const int ch = v * 64 + lane;     // 512 chunks of 8 columns: 32 rows x 16 chunks
            const int rr = ch >> 4, cc = (ch & 15) * 8;
            const long r = m0 + wave * 32 + rr;
            const int n = n0 + cc;
            const uint4 o = *reinterpret_cast<const uint4 *>(&stage[rr * NT_OUT_LD + cc]);
            // Unconditional buffer store; lanes outside the matrix get an out-of-range offset, which
            // the hardware drops.  A store under a branch would hide the number of stores in flight
            // from the compiler, and the wait for the prefetched operands at the top of the next
            // stage would become a wait for every store acknowledgement (vmcnt(0)).
            const int off = (r < R && n < N) ? (int)(((long)(wave * 32 + rr) * N + n) * 2) : -1;
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{o.x, o.y, o.z, o.w}, orsrc, off, 0, 0);
            if (RED) {
                float dz[8], yv[8];
                unpack8(o, dz);
                unpack8(yraw[v], yv);
                const int c8 = (lane & 15) * 8;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float4 sc = *reinterpret_cast<const float4 *>(&rconst[0 * NT_BN + c8 + 4 * h]);
                    const float4 sh = *reinterpret_cast<const float4 *>(&rconst[1 * NT_BN + c8 + 4 * h]);
                    const float4 mu = *reinterpret_cast<const float4 *>(&rconst[2 * NT_BN + c8 + 4 * h]);
                    const float4 is = *reinterpret_cast<const float4 *>(&rconst[3 * NT_BN + c8 + 4 * h]);
                    const float rsc[4] = {sc.x, sc.y, sc.z, sc.w}, rsh[4] = {sh.x, sh.y, sh.z, sh.w};
                    const float rmu[4] = {mu.x, mu.y, mu.z, mu.w}, ris[4] = {is.x, is.y, is.z, is.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int e = 4 * h + i;
                        const float du = dz[e] * act_grad(fmaf(yv[e], rsc[i], rsh[i]), red_slope);
                        rs1[e] += du;
                        rs2[e] = fmaf(du, (yv[e] - rmu[i]) * ris[i], rs2[e]);
                    }
                }
            }
        }
        if (RED) {
            // lanes l, l+16, l+32, l+48 share their columns; lanes 0..15 add into the wave's slab
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                rs1[i] += __shfl_xor(rs1[i], 16);
                rs1[i] += __shfl_xor(rs1[i], 32);
                rs2[i] += __shfl_xor(rs2[i], 16);
                rs2[i] += __shfl_xor(rs2[i], 32);
            }
            if (lane < 16) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    ssum[(wave * 2 + 0) * NT_BN + lane * 8 + i] += rs1[i];
                    ssum[(wave * 2 + 1) * NT_BN + lane * 8 + i] += rs2[i];
                }
            }
        }
        __syncthreads();  // the next stage's LDS writes must not overtake another wave's read-back
    }
    if (STATS || RED) {
        // lanes l and l+32 hold the same column (different rows); then the 4 waves meet in LDS; the
        // workgroup's totals go to ITS slot of the partials buffer (no atomics: pcb_bn_finalize /
        // pcb_bn_bwd_finalize sum the slots in a fixed order)
#pragma unroll
        for (int j = 0; j < (STATS ? 4 : 0); ++j) {
            const float s2 = st_s[j] + __shfl_xor(st_s[j], 32);
            const float q2 = st_q[j] + __shfl_xor(st_q[j], 32);
            if (lane < 32) {
                ssum[(wave * 2 + 0) * NT_BN + j * 32 + lane] = s2;
                ssum[(wave * 2 + 1) * NT_BN + j * 32 + lane] = q2;
            }
        }
        __syncthreads();
        if (t < NT_BN && n0 + t < N) {
            float a = 0.0f, b = 0.0f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                a += ssum[(w * 2 + 0) * NT_BN + t];
                b += ssum[(w * 2 + 1) * NT_BN + t];
            }
            if (STATS && red_arg.centre && (long)blockIdx.x == (tiles_m - 1) % (long)gridDim.x) {
                // rows past R: their accumulators are exact zeros, so the centred epilogue turned each of them into
                // bf16(-centre) (never stored: the buffer store drops them) -- taken out of the statistics again here, by
                // the workgroup that handled the last row tile, instead of a select per element in every tile
                const float pad = (float)(tiles_m * NT_BM - R), v = bf2f(f2bf(-red_arg.centre[n0 + t]));
                a -= pad * v;
                b -= pad * (v * v);
            }
            sums[((long)blockIdx.x * 2 + 0) * N + n0 + t] = a;
            sums[((long)blockIdx.x * 2 + 1) * N + n0 + t] = b;
        }
    }
}

// ---- gemm_nt, eight-wave form -------------------------------------------------------------------
// The 128 x 128 x 64 stages of gemm_nt_kernel run by 512 threads (4 x 2 waves, 32 rows x 64 columns
// each), for the forward prologues (plain, BatchNorm+activation on load).  At the 168 registers
// that three four-wave workgroups per CU leave a thread, the compiler reads every B fragment of the
// four-wave kernel into the same four registers -- each MFMA waits for its own LDS read -- and the
// per-column constants, fetched from global memory at the top of every stage, put an L2 round trip
// on each stage's critical path.  Here a wave stages half as many chunks and holds half the
// accumulators (fragments are double-buffered within 128 registers, 16 waves per CU), and the
// constants come from LDS (all K columns, loaded once per workgroup).
// Measured on the shapes of a pn2_msg step (tools/nt_bench.py): forward launches 5-25 % shorter.
// What was tried and is NOT here: a second register set of staged operands (spills at any
// occupancy that keeps 16 waves), 64-row tiles at 3 or 4 workgroups per CU (slower: twice the
// weight traffic from L2).
constexpr int N8_MAXK = 512;  // columns of constants held in LDS; longer rows use gemm_nt_kernel

template <int PRO>
struct ConstsLds {
    float scale[8], shift[8], p[8], q[8];
    // cst = [4][kpad] floats in LDS: scale | shift | p | q
    __device__ __forceinline__ void load(const float *cst, int kpad, int c)
    {
        if (PRO == PRO_PLAIN) return;
        rd8(cst + c, scale);
        rd8(cst + kpad + c, shift);
        if (PRO >= PRO_DY) {
            rd8(cst + 2 * kpad + c, p);
            rd8(cst + 3 * kpad + c, q);
        }
    }
    static __device__ __forceinline__ void rd8(const float *src, float *dst)
    {
        const float4 a = *reinterpret_cast<const float4 *>(src);
        const float4 b = *reinterpret_cast<const float4 *>(src + 4);
        dst[0] = a.x; dst[1] = a.y; dst[2] = a.z; dst[3] = a.w;
        dst[4] = b.x; dst[5] = b.y; dst[6] = b.z; dst[7] = b.w;
    }
};

// Raw<PRO>::finish with constants given as a ConstsLds (same arithmetic, same order)
template <int PRO>
__device__ __forceinline__ uint4 finish_with(const Raw<PRO> &r, const ConstsLds<PRO> &k, float slope)
{
    Consts<PRO> c;
    if (PRO != PRO_PLAIN) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            c.scale[i] = k.scale[i];
            c.shift[i] = k.shift[i];
            if (PRO >= PRO_DY) {
                c.p[i] = k.p[i];
                c.q[i] = k.q[i];
            }
        }
    }
    return r.finish(c, slope);
}

template <int PRO, int NA, int NB>
struct Slot8 {
    Raw<PRO> ra[NA];
    uint4 rb[NB];
    uint32_t keepb[NB];
};

template <int PRO, int STATS, int RED, int WM, int OCC>
__global__ __launch_bounds__(WM * 128, OCC) void gemm_nt8_kernel(Operand A_arg, const u16 *__restrict__ Bw, long R,
                                                                           int N, int K, u16 *__restrict__ out,
                                                                           float *__restrict__ sums, RedArgs red_arg)
{
    const Operand A = local_copy(A_arg);
    const float a_slope = act_slope(A.act), red_slope = act_slope(red_arg.act);
    RedArgs red;
    red.y = red_arg.y; red.scale = red_arg.scale; red.shift = red_arg.shift;
    red.mean = red_arg.mean; red.invstd = red_arg.invstd; red.act = red_arg.act;
    // [A stage | B stage] in the main loop, per-wave 32 x 64 output staging in the epilogue (same size)
    constexpr int BM = WM * 32, THREADS = WM * 128, AROWS = THREADS / 8, NA = BM / AROWS, NB = NT_BN / AROWS;
    __shared__ __attribute__((aligned(16))) u16 smem[(BM + NT_BN) * NT_LD];
    static_assert(2 * WM * 32 * (64 + 8) <= (BM + NT_BN) * NT_LD, "output staging must fit the stage buffers");
    u16 *const As = smem;
    u16 *const Bs = smem + BM * NT_LD;
    __shared__ float ssum[WM * 2 * NT_BN];                                            // [wm][s|q][column]
    __shared__ __attribute__((aligned(16))) float rconst[RED ? 4 * NT_BN : 4];      // RED: constants of the layer below
    __shared__ __attribute__((aligned(16))) float cst[PRO == PRO_PLAIN ? 4 : (PRO == PRO_BNACT ? 2 : 4) * N8_MAXK];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int n0 = blockIdx.y * NT_BN;
    const int chunk = t & 7;   // 8-column chunk inside a BK stage
    const int rrow = t >> 3;   // 0..AROWS-1
    const long tiles_m = (R + BM - 1) / BM;
    constexpr int OLD = 64 + 8;  // output staging row stride

    long tile = blockIdx.x;
    if (tile >= tiles_m) {
        if ((STATS || RED) && t < NT_BN && n0 + t < N) {
            sums[((long)blockIdx.x * 2 + 0) * N + n0 + t] = 0.0f;
            sums[((long)blockIdx.x * 2 + 1) * N + n0 + t] = 0.0f;
        }
        return;
    }
    const int kpad = K;  // cst row length (K % 8 == 0)
    if (PRO != PRO_PLAIN) {
        for (int i = t; i < K; i += THREADS) {
            cst[i] = A.scale[i];
            cst[kpad + i] = A.shift[i];
            if (PRO >= PRO_DY) {
                cst[2 * kpad + i] = A.p[i];
                cst[3 * kpad + i] = A.q[i];
            }
        }
    }
    if (STATS || RED)
        for (int i = t; i < WM * 2 * NT_BN; i += THREADS) ssum[i] = 0.0f;
    if (RED && t < NT_BN) {
        const bool ok = n0 + t < N;
        rconst[0 * NT_BN + t] = ok ? red.scale[n0 + t] : 0.0f;
        rconst[1 * NT_BN + t] = ok ? red.shift[n0 + t] : 0.0f;
        rconst[2 * NT_BN + t] = ok ? red.mean[n0 + t] : 0.0f;
        rconst[3 * NT_BN + t] = ok ? red.invstd[n0 + t] : 0.0f;
    }

    f32x16 acc[2];
    float st_s[2] = {0.0f, 0.0f}, st_q[2] = {0.0f, 0.0f};

    Slot8<PRO, NA, NB> s0;
    long ft = tile;  // the (tile, k-stage) pair the next fetch loads
    int fk = 0;
    auto fetch = [&](Slot8<PRO, NA, NB> &s) {
        if (ft < tiles_m) {
            const int kc = fk + chunk * 8;
            const long mb = ft * BM;
#pragma unroll
            for (int i = 0; i < NA; ++i) s.ra[i].load(A, mb + rrow + AROWS * i, kc, R, K);
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int n = n0 + rrow + AROWS * i;
                s.rb[i] = *reinterpret_cast<const uint4 *>(Bw + (long)(n < N ? n : N - 1) * K + (kc < K ? kc : 0));
                s.keepb[i] = (n < N && kc < K) ? 0xffffffffu : 0u;
            }
        }
        fk += NT_BK;
        if (fk >= K) {
            fk = 0;
            ft += gridDim.x;
        }
    };
    fetch(s0);
    __syncthreads();  // cst, ssum, rconst

    auto stage = [&](Slot8<PRO, NA, NB> &s, int k0) {
        ConstsLds<PRO> ka;
        ka.load(cst, kpad, (k0 + chunk * 8 < K) ? k0 + chunk * 8 : 0);
#pragma unroll
        for (int i = 0; i < NA; ++i)
            *reinterpret_cast<uint4 *>(&As[(rrow + AROWS * i) * NT_LD + chunk * 8]) = finish_with<PRO>(s.ra[i], ka, a_slope);
#pragma unroll
        for (int i = 0; i < NB; ++i)
            *reinterpret_cast<uint4 *>(&Bs[(rrow + AROWS * i) * NT_LD + chunk * 8]) =
                make_uint4(s.rb[i].x & s.keepb[i], s.rb[i].y & s.keepb[i], s.rb[i].z & s.keepb[i], s.rb[i].w & s.keepb[i]);
        __syncthreads();
        fetch(s);  // the registers just emptied take the next stage: its loads fly under the MFMAs below
        bf16x8 fa[2], fb[2][2];
        auto frags = [&](int buf, int ks) {
            const int kk = ks * 16 + (lane >> 5) * 8;
            fa[buf] = *reinterpret_cast<const bf16x8 *>(&As[(wm * 32 + (lane & 31)) * NT_LD + kk]);
#pragma unroll
            for (int j = 0; j < 2; ++j)
                fb[buf][j] = *reinterpret_cast<const bf16x8 *>(&Bs[(wn * 64 + j * 32 + (lane & 31)) * NT_LD + kk]);
        };
        frags(0, 0);
#pragma unroll
        for (int ks = 0; ks < NT_BK / 16; ++ks) {
            if (ks + 1 < NT_BK / 16) frags((ks + 1) & 1, ks + 1);
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks & 1], fb[ks & 1][j], acc[j], 0, 0, 0);
        }
        __syncthreads();
    };

    for (; tile < tiles_m; tile += gridDim.x) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[j][i] = 0.0f;
        for (int k0 = 0; k0 < K; k0 += NT_BK) stage(s0, k0);
        const long m0 = tile * BM;
        u16 *const stg = smem + wave * (32 * OLD);
        const long rows_here = R - m0 < BM ? R - m0 : BM;
        const __amdgpu_buffer_rsrc_t orsrc =
            __builtin_amdgcn_make_buffer_rsrc(out + m0 * N, 0, (int)(rows_here * N * 2), 0x00020000);
        uint4 yraw[RED ? 4 : 1];
        if (RED) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int ch = v * 64 + lane;  // 256 chunks: 32 rows x 8 chunks of 8 columns
                const long r = m0 + wm * 32 + (ch >> 3);
                const int n = n0 + wn * 64 + (ch & 7) * 8;
                yraw[v] = *reinterpret_cast<const uint4 *>(red.y + (r < R ? r : R - 1) * N + (n < N ? n : 0));
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float sv = 0.0f, sq = 0.0f;
            float bj = 0.0f;
            if (!STATS && !RED && red_arg.bias) {
                const int n = n0 + wn * 64 + j * 32 + (lane & 31);
                bj = n < N ? red_arg.bias[n] : 0.0f;
            }
            if (STATS && red_arg.centre) {   // (wave-uniform) rows stored centred: see RedArgs
                const int n = n0 + wn * 64 + j * 32 + (lane & 31);
                bj = n < N ? -red_arg.centre[n] : 0.0f;
            }
            float ad[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if (STATS == 2) {
                // the lane's 16 accumulators are 4 runs of 4 consecutive rows (aligned to 4): one coarse row each
                // (sh >= 2).  Rows past R take nothing: they must stay out of the statistics.
                const int n = n0 + wn * 64 + j * 32 + (lane & 31);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const long r = m0 + wm * 32 + 8 * g + 4 * (lane >> 5);
                    if (r < R && n < N) {
                        ad[g] = red_arg.add1[(r >> red_arg.sh1) * N + n];
                        if (red_arg.add2) ad[g] += red_arg.add2[(r >> red_arg.sh2) * N + n];
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int rr = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                const u16 h = f2bf(RED ? acc[j][i] : (STATS == 2 ? (acc[j][i] + ad[i >> 2]) + bj : acc[j][i] + bj));
                stg[rr * OLD + j * 32 + (lane & 31)] = h;
                if (STATS) {
                    const float v = bf2f(h);
                    sv += v;
                    sq = fmaf(v, v, sq);
                }
            }
            if (STATS) {
                st_s[j] += sv;
                st_q[j] += sq;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
        float rs1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, rs2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int ch = v * 64 + lane;
            const int rr = ch >> 3, cc = (ch & 7) * 8;
            const long r = m0 + wm * 32 + rr;
            const int n = n0 + wn * 64 + cc;
            uint4 o = *reinterpret_cast<const uint4 *>(&stg[rr * OLD + cc]);
            const int off = (r < R && n < N) ? (int)(((long)(wm * 32 + rr) * N + n) * 2) : -1;
            if (!STATS && !RED && red_arg.res) {   // (wave-uniform) the other branch's rows, added to the rounded result
                const uint4 rv = *reinterpret_cast<const uint4 *>(red_arg.res + (r < R ? r : R - 1) * N + (n < N ? n : 0));
                float fo[8], fr[8];
                unpack8(o, fo);
                unpack8(rv, fr);
#pragma unroll
                for (int i = 0; i < 8; ++i) fo[i] += fr[i];
                o = pack8(fo);
            }
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{o.x, o.y, o.z, o.w}, orsrc, off, 0, 0);
            if (RED) {
                float dz[8], yv[8];
                unpack8(o, dz);
                unpack8(yraw[v], yv);
                const int c8 = wn * 64 + (lane & 7) * 8;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float4 sc = *reinterpret_cast<const float4 *>(&rconst[0 * NT_BN + c8 + 4 * h]);
                    const float4 sh = *reinterpret_cast<const float4 *>(&rconst[1 * NT_BN + c8 + 4 * h]);
                    const float4 mu = *reinterpret_cast<const float4 *>(&rconst[2 * NT_BN + c8 + 4 * h]);
                    const float4 is = *reinterpret_cast<const float4 *>(&rconst[3 * NT_BN + c8 + 4 * h]);
                    const float rsc[4] = {sc.x, sc.y, sc.z, sc.w}, rsh[4] = {sh.x, sh.y, sh.z, sh.w};
                    const float rmu[4] = {mu.x, mu.y, mu.z, mu.w}, ris[4] = {is.x, is.y, is.z, is.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int e = 4 * h + i;
                        const float du = dz[e] * act_grad(fmaf(yv[e], rsc[i], rsh[i]), red_slope);
                        rs1[e] += du;
                        rs2[e] = fmaf(du, (yv[e] - rmu[i]) * ris[i], rs2[e]);
                    }
                }
            }
        }
        if (RED) {
            // lanes l, l+8, ..., l+56 share their 8 columns; lanes 0..7 add into the wave's slab
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                rs1[i] += __shfl_xor(rs1[i], 8);
                rs1[i] += __shfl_xor(rs1[i], 16);
                rs1[i] += __shfl_xor(rs1[i], 32);
                rs2[i] += __shfl_xor(rs2[i], 8);
                rs2[i] += __shfl_xor(rs2[i], 16);
                rs2[i] += __shfl_xor(rs2[i], 32);
            }
            if (lane < 8) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    ssum[(wm * 2 + 0) * NT_BN + wn * 64 + lane * 8 + i] += rs1[i];
                    ssum[(wm * 2 + 1) * NT_BN + wn * 64 + lane * 8 + i] += rs2[i];
                }
            }
        }
        __syncthreads();  // the next stage's LDS writes must not overtake another wave's read-back
    }
    if (STATS || RED) {
#pragma unroll
        for (int j = 0; j < (STATS ? 2 : 0); ++j) {
            const float s2 = st_s[j] + __shfl_xor(st_s[j], 32);
            const float q2 = st_q[j] + __shfl_xor(st_q[j], 32);
            if (lane < 32) {
                ssum[(wm * 2 + 0) * NT_BN + wn * 64 + j * 32 + lane] = s2;
                ssum[(wm * 2 + 1) * NT_BN + wn * 64 + j * 32 + lane] = q2;
            }
        }
        __syncthreads();
        if (t < NT_BN && n0 + t < N) {
            float a = 0.0f, b = 0.0f;
#pragma unroll
            for (int w = 0; w < WM; ++w) {
                a += ssum[(w * 2 + 0) * NT_BN + t];
                b += ssum[(w * 2 + 1) * NT_BN + t];
            }
            if (STATS && red_arg.centre && (long)blockIdx.x == (tiles_m - 1) % (long)gridDim.x) {
                // rows past R took bf16(-centre) in the centred epilogue (see gemm_nt_kernel): out of the statistics again
                const float pad = (float)(tiles_m * BM - R), v = bf2f(f2bf(-red_arg.centre[n0 + t]));
                a -= pad * v;
                b -= pad * (v * v);
            }
            sums[((long)blockIdx.x * 2 + 0) * N + n0 + t] = a;
            sums[((long)blockIdx.x * 2 + 1) * N + n0 + t] = b;
        }
    }
}

// ---- gemm_nt with the transformed A tile resident in LDS -------------------------------------
// For the input-gradient GEMMs whose output is wider than one 128-column tile (N = 264: the layers
// fed by grouped / concatenated rows) the plain kernel would rebuild A' = dy(dz, y) once per column
// tile.  Here a workgroup builds the 64 x K tile of A' once (K <= 256), keeps it in LDS and walks
// the column tiles itself; only the weight stages (L2-resident) move through the register pipeline.
constexpr int AR_BM = 64, AR_BN = 128, AR_BK = 64;
constexpr int AR_BLD = AR_BK + 8;
constexpr int AR_OLD = 64 + 8;  // per-wave output staging: 32 rows x 64 columns

template <int PRO, int KCH>  // KCH = K/8 chunks per row held in LDS: 16 (K <= 128) or 32 (K <= 256)
__global__ __launch_bounds__(256, 3) void gemm_nt_ares_kernel(Operand A_arg, const u16 *__restrict__ Bw, long R,
                                                            int N, int K, u16 *__restrict__ out)
{
    const Operand A = local_copy(A_arg);
    const float a_slope = act_slope(A.act);
    constexpr int ALD = KCH * 8 + 8;      // A row stride in bf16: rows 16 B apart in bank space
    constexpr int RPT = 256 / KCH;        // rows covered by one pass of the 256 threads
    __shared__ __attribute__((aligned(16))) u16 As[AR_BM * ALD];
    __shared__ __attribute__((aligned(16))) u16 Bs[AR_BN * AR_BLD];
    // the per-wave output staging blocks share the weight stage buffer (idle between a column tile's last barrier and the next
    // tile's first stage; one more barrier per column tile): 52 instead of 71 KB at K = 256 -- three workgroups per CU
    static_assert(4 * 32 * AR_OLD <= AR_BN * AR_BLD, "output staging must fit the weight stage");
    u16 *const Os = Bs;

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;  // 2 x 2 waves: 32 rows x 64 columns each
    const long tiles_m = (R + AR_BM - 1) / AR_BM;
    const int n_tiles = (N + AR_BN - 1) / AR_BN;
    const int bchunk = t & 7, brow = t >> 3;

    uint4 rb[4];
    uint32_t keepb[4];
    auto fetch_b = [&](int nt, int k0) {
        const int kc = k0 + bchunk * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = nt * AR_BN + brow + 32 * i;
            rb[i] = *reinterpret_cast<const uint4 *>(Bw + (long)(n < N ? n : N - 1) * K + (kc < K ? kc : 0));
            keepb[i] = (n < N && kc < K) ? 0xffffffffu : 0u;
        }
    };

    for (long tile = blockIdx.x; tile < tiles_m; tile += gridDim.x) {
        const long m0 = tile * AR_BM;
        {   // phase 1: the whole transformed A tile, once
            const int kc = (t % KCH) * 8;
            Consts<PRO> ka;
            ka.load(A, kc, K);
            Raw<PRO> ra[AR_BM / RPT];
#pragma unroll
            for (int i = 0; i < AR_BM / RPT; ++i) ra[i].load(A, m0 + t / KCH + RPT * i, kc, R, K);
#pragma unroll
            for (int i = 0; i < AR_BM / RPT; ++i)
                *reinterpret_cast<uint4 *>(&As[(t / KCH + RPT * i) * ALD + kc]) = ra[i].finish(ka, a_slope);
        }
        fetch_b(0, 0);
        __syncthreads();
        for (int nt = 0; nt < n_tiles; ++nt) {
            f32x16 acc[2];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[j][i] = 0.0f;
            for (int k0 = 0; k0 < K; k0 += AR_BK) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    *reinterpret_cast<uint4 *>(&Bs[(brow + 32 * i) * AR_BLD + bchunk * 8]) =
                        make_uint4(rb[i].x & keepb[i], rb[i].y & keepb[i], rb[i].z & keepb[i], rb[i].w & keepb[i]);
                __syncthreads();
                const bool last_k = k0 + AR_BK >= K;
                const int nnt = last_k ? nt + 1 : nt;
                if (nnt < n_tiles) fetch_b(nnt, last_k ? 0 : k0 + AR_BK);
#pragma unroll
                for (int ks = 0; ks < AR_BK / 16; ++ks) {
                    const int kk = ks * 16 + (lane >> 5) * 8;
                    const bf16x8 a = *reinterpret_cast<const bf16x8 *>(&As[(wm * 32 + (lane & 31)) * ALD + k0 + kk]);
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const bf16x8 b = *reinterpret_cast<const bf16x8 *>(&Bs[(wn * 64 + j * 32 + (lane & 31)) * AR_BLD + kk]);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
                    }
                }
                __syncthreads();
            }
            // column tile done: 16-byte row-segment stores through the wave's staging block
            u16 *const stage = Os + wave * (32 * AR_OLD);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    stage[((i & 3) + 8 * (i >> 2) + 4 * (lane >> 5)) * AR_OLD + j * 32 + (lane & 31)] = f2bf(acc[j][i]);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int ch = v * 64 + lane;  // 256 chunks: 32 rows x 8 chunks of 8 columns
                const int rr = ch >> 3, cc = (ch & 7) * 8;
                const long r = m0 + wm * 32 + rr;
                const int n = nt * AR_BN + wn * 64 + cc;
                if (r < R && n < N)
                    *reinterpret_cast<uint4 *>(out + r * N + n) = *reinterpret_cast<const uint4 *>(&stage[rr * AR_OLD + cc]);
            }
            __syncthreads();  // the staging blocks are the weight stage buffer: read back before the next stage is parked
        }
        // (As is rebuilt for the next row tile: the barrier above covers it)
    }
}

// ---- gemm_tn (weight gradient) ----------------------------------------------------------------
constexpr int TN_BM = 128, TN_BN = 128, TN_RS = 32;  // RS rows of the reduction per stage
constexpr int TN_LD = 128 + 32;                       // row stride 320 B: conflict-free tr reads

// Workgroup id -> position in a logical order in which neighbours share an XCD.  The dispatcher
// deals consecutive workgroup ids round-robin over the 8 XCDs, each with its own L2; tiles that
// read the same operand rows should therefore sit 8 ids apart, not next to each other.  This
// lists XCD 0's workgroups first, then XCD 1's, ... (a bijection for any grid size).
__device__ __forceinline__ unsigned xcd_logical(unsigned id, unsigned total)
{
    const unsigned xcd = id & 7u, slot = id >> 3;
    const unsigned q = total >> 3, rem = total & 7u;
    return xcd * q + (xcd < rem ? xcd : rem) + slot;
}

// ASUM (plain A operand only): the column sums of A over the split's rows are stored behind the split's [M,N]
// slab -- for a conv with a bias that is its bias gradient (sum of dy over the rows), which otherwise costs a
// pass of its own over dy.  `stride` = floats from one split's slab to the next (M*N, or M*N + M with ASUM).
// TEAMS = 2: two four-wave teams in one workgroup, each with its own stage buffers and its own half of the split's
// rows; at the end team 1 hands its accumulators to team 0 through LDS and ONE slab is written for both.  The slabs
// (splits x M x N floats, written here and read again by the slab sum) were a third of the traffic of the step's
// weight gradients at one slab per four waves; the waves per CU, their loads and MFMAs are the same as before.
// TWA / TWB = 64: an operand of at most 64 columns (the first layers of the set-abstraction stacks, the narrow attention
// layers) gets a 64-wide tile -- 32 columns per wave -- and the stage grows to 64 rows: a 128-wide tile would stage and
// multiply half (both narrow: three quarters) of padding for it.
template <int APRO, int BPRO, int ASUM = 0, int TEAMS = 1, int TWA = 128, int TWB = TWA>
__global__ __launch_bounds__(256 * TEAMS, TEAMS == 1 ? 2 : 1) void gemm_tn_kernel(Operand A_arg, Operand B_arg, long R, int M, int N,
                                                       long rows_per_split, float *__restrict__ part, long stride,
                                                       int tiles_m, int tiles_n)
{
    static_assert(!ASUM || APRO == PRO_PLAIN, "column sums of the plain operand only");
    const Operand A = local_copy(A_arg);
    const Operand B = local_copy(B_arg);
    const float a_slope = act_slope(A.act), b_slope = act_slope(B.act);
    constexpr int RS = (TWA == 64 || TWB == 64) ? 64 : TN_RS;   // rows of the reduction per stage
    constexpr int LDA = TWA + 32, LDB = TWB + 32;  // LDS row strides in bf16 (320 / 192 B: conflict-free tr reads)
    constexpr int CHA = TWA / 8, RPPA = 256 / CHA; // 16-byte chunks per row; rows one pass of the team's 256 threads covers
    constexpr int CHB = TWB / 8, RPPB = 256 / CHB;
    constexpr int NCHA = RS / RPPA, NCHB = RS / RPPB;  // chunks per thread and stage
    constexpr int XNA = TWA / 64, XNB = TWB / 64;  // 32-wide MFMA tiles per wave
    static_assert(TWA == 128 ? LDA == TN_LD : true, "the 128-wide form keeps its layout");
    static_assert(TEAMS == 1 || (XNA * XNB * 16 + 8) * 256 * 4 <= TEAMS * 2 * RS * (LDA + LDB) * 2, "the exchange must fit the stage buffers");
    // per team: two stage buffers per operand
    __shared__ __attribute__((aligned(16))) u16 smem[TEAMS * 2 * RS * (LDA + LDB)];
    const int team = TEAMS == 1 ? 0 : (int)(threadIdx.x >> 8);
    u16 *const As = smem + team * (2 * RS * (LDA + LDB));
    u16 *const Bs = As + 2 * RS * LDA;

    const int t = threadIdx.x & 255;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;  // 2 x 2 waves, (TWA / 2) x (TWB / 2) outputs each
    // 1-D grid; the N tiles (then M tiles) of one row split are neighbours on one XCD, so the
    // operand rows they share are fetched from HBM once and found in that XCD's L2 afterwards
    const unsigned logical = xcd_logical(blockIdx.x, gridDim.x);
    const int tile_n = logical % tiles_n;
    const int tile_m = (logical / tiles_n) % tiles_m;
    const int split = logical / (tiles_n * tiles_m);
    const int m0 = tile_m * TWA;
    const int n0 = tile_n * TWB;
    // the workgroup's rows, divided among its teams in whole stages; every team runs the same number of stages
    // (the barriers are the workgroup's): rows past a team's range load as zeros
    const long wg_begin = (long)split * rows_per_split;
    const long wg_end = wg_begin + rows_per_split < R ? wg_begin + rows_per_split : R;
    const long per_team = ((wg_end - wg_begin + TEAMS - 1) / TEAMS + RS - 1) / RS * RS;
    const long r_begin = wg_begin + team * per_team < wg_end ? wg_begin + team * per_team : wg_end;
    const long r_end = r_begin + per_team < wg_end ? r_begin + per_team : wg_end;
    const long r_stop = r_begin + (wg_begin < wg_end ? per_team : 0);   // uniform stage count

    f32x16 acc[XNA][XNB];
#pragma unroll
    for (int a = 0; a < XNA; ++a)
#pragma unroll
        for (int b = 0; b < XNB; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.0f;

    // staging: RS rows x TW / 8 chunks of 8 columns per operand -> NCHA / NCHB chunks per thread.
    // A thread's column chunk never changes, so its per-column constants stay in registers.
    const int chunk = t % CHA, chunkb = t % CHB;
    const int rrow = t / CHA, rrowb = t / CHB;  // 0..RPP-1
    Consts<APRO> ka;
    Consts<BPRO> kb;
    ka.load(A, m0 + chunk * 8, M);
    kb.load(B, n0 + chunkb * 8, N);
    Raw<APRO> ra[NCHA];
    Raw<BPRO> rb[NCHB];
    auto fetch = [&](long r0) {
#pragma unroll
        for (int i = 0; i < (NCHA > NCHB ? NCHA : NCHB); ++i) {   // (interleaved: the order the 128-wide form was tuned with)
            if (i < NCHA) {
                const long r = r0 + rrow + RPPA * i;
                ra[i].load(A, r < r_end ? r : R, m0 + chunk * 8, R, M);
            }
            if (i < NCHB) {
                const long r = r0 + rrowb + RPPB * i;
                rb[i].load(B, r < r_end ? r : R, n0 + chunkb * 8, R, N);
            }
        }
    };
    // transposed-read addresses: lane 4q+p of a 16-lane group supplies row q, columns 4p..4p+3 of a
    // 4 x 16 block; group g covers columns 16*(g&1).. of a 32-wide MFMA tile and k rows 8*(g>>1)..
    const int grp = lane >> 4, gi = lane & 15;
    const int tr_row = 8 * (grp >> 1) + (gi >> 2);
    const int tr_col = 16 * (grp & 1) + 4 * (gi & 3);

    // Two LDS stage buffers, ONE barrier per stage: while the waves multiply stage s out of buffer s & 1, the
    // registers hold stage s+1 (its loads were issued before the previous barrier); it is transformed and parked in
    // the other buffer behind the MFMAs, the loads of stage s+2 go out, and the barrier at the end of the step both
    // publishes buffer (s+1) & 1 and retires the reads of buffer s & 1.
    float asum[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    auto park = [&](int buf) {
        u16 *const Ab = As + buf * (RS * LDA);
        u16 *const Bb = Bs + buf * (RS * LDB);
#pragma unroll
        for (int i = 0; i < (NCHA > NCHB ? NCHA : NCHB); ++i) {
            if (i < NCHA) {
                const uint4 av = ra[i].finish(ka, a_slope);
                *reinterpret_cast<uint4 *>(&Ab[(rrow + RPPA * i) * LDA + chunk * 8]) = av;
                if (ASUM) {
                    float f[8];
                    unpack8(av, f);  // rows outside the split are zero already
#pragma unroll
                    for (int e = 0; e < 8; ++e) asum[e] += f[e];
                }
            }
            if (i < NCHB) *reinterpret_cast<uint4 *>(&Bb[(rrowb + RPPB * i) * LDB + chunkb * 8]) = rb[i].finish(kb, b_slope);
        }
    };
    if (r_begin < r_stop) {
        fetch(r_begin);
        park(0);
        if (r_begin + RS < r_stop) fetch(r_begin + RS);
    }
    __syncthreads();
    int cur = 0;
    for (long r0 = r_begin; r0 < r_stop; r0 += RS, cur ^= 1) {
        const u16 *const Ab = As + cur * (RS * LDA);
        const u16 *const Bb = Bs + cur * (RS * LDB);
#pragma unroll
        for (int ks = 0; ks < RS / 16; ++ks) {
            bf16x8 af[XNA], bf[XNB];
            typedef __attribute__((address_space(3))) s16x4 *lds_ptr;
#pragma unroll
            for (int x = 0; x < (XNA > XNB ? XNA : XNB); ++x) {
                if (x < XNA) {
                    const u16 *pa = &Ab[(ks * 16 + tr_row) * LDA + wm * (TWA / 2) + x * 32 + tr_col];
                    const s16x4 a_lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)pa);
                    const s16x4 a_hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(pa + 4 * LDA));
                    af[x] = __builtin_shufflevector(a_lo, a_hi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
                if (x < XNB) {
                    const u16 *pb = &Bb[(ks * 16 + tr_row) * LDB + wn * (TWB / 2) + x * 32 + tr_col];
                    const s16x4 b_lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)pb);
                    const s16x4 b_hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(pb + 4 * LDB));
                    bf[x] = __builtin_shufflevector(b_lo, b_hi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
            }
#pragma unroll
            for (int a = 0; a < XNA; ++a)
#pragma unroll
                for (int b = 0; b < XNB; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
        if (r0 + RS < r_stop) {
            park(cur ^ 1);
            if (r0 + 2 * RS < r_stop) fetch(r0 + 2 * RS);
        }
        __syncthreads();
    }

    if (TEAMS == 2) {
        // team 1's accumulators (and column sums) join team 0's: thread t of both teams holds the same outputs, so the
        // exchange is [value][t] (conflict-free); 64 + 8 values x 256 floats = 72 KB of the 80 KB of stage buffers
        float *const xch = reinterpret_cast<float *>(smem);
        if (team == 1) {
#pragma unroll
            for (int a = 0; a < XNA; ++a)
#pragma unroll
                for (int b = 0; b < XNB; ++b)
#pragma unroll
                    for (int i = 0; i < 16; ++i) xch[((a * XNB + b) * 16 + i) * 256 + t] = acc[a][b][i];
            if (ASUM)
#pragma unroll
                for (int e = 0; e < 8; ++e) xch[(XNA * XNB * 16 + e) * 256 + t] = asum[e];
        }
        __syncthreads();
        if (team == 0) {
#pragma unroll
            for (int a = 0; a < XNA; ++a)
#pragma unroll
                for (int b = 0; b < XNB; ++b)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[a][b][i] += xch[((a * XNB + b) * 16 + i) * 256 + t];
            if (ASUM)
#pragma unroll
                for (int e = 0; e < 8; ++e) asum[e] += xch[(XNA * XNB * 16 + e) * 256 + t];
        }
    }
    if (team == 0) {
#pragma unroll
    for (int a = 0; a < XNA; ++a)
#pragma unroll
        for (int b = 0; b < XNB; ++b) {
            const int n = n0 + wn * (TWB / 2) + b * 32 + (lane & 31);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int m = m0 + wm * (TWA / 2) + a * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                // one plain store per element into this split's slab (summed by reduce_slabs_kernel):
                // thousands of workgroups adding into one small dW would serialise on its few lines
                if (m < M && n < N) part[(long)split * stride + (long)m * N + n] = acc[a][b][i];
            }
        }
    }
    if (ASUM && tile_n == 0) {
        // the 16 threads that share a column chunk (t & 15) meet in LDS (the stage buffers are idle now), fixed order
        float *const red = reinterpret_cast<float *>(smem);   // [RPPA][TWA] (team 0's values)
        __syncthreads();
        if (team == 0)
#pragma unroll
        for (int e = 0; e < 8; ++e) red[rrow * TWA + chunk * 8 + e] = asum[e];
        __syncthreads();
        if (team == 0 && t < TWA && m0 + t < M) {
            float a = 0.0f;
#pragma unroll
            for (int i = 0; i < RPPA; ++i) a += red[i * TWA + t];
            part[(long)split * stride + (long)M * N + m0 + t] = a;
        }
    }
}

// ---- one layer's backward in ONE pass: input gradient + weight gradient + the sums of the layer below ----------------
// The backward of layer l of a stack is two GEMMs over the same operand: dx = dy W (gemm_nt with the dy prologue and the
// RED epilogue) and dW = dy^T x' (gemm_tn with the same prologue).  Run as two kernels, (dz, y) of the layer stream from HBM
// twice, y of the layer below twice as well (x' for dW, the RED epilogue), and the 15 vector instructions per MFMA of the dy
// prologue are paid twice.  For the NARROW layers that carry most of the step's rows (the set-abstraction stacks: C, K <= 128
// over 131072 .. 524288 rows) everything fits one workgroup: a 64-row tile of dy [64, C] and of x' [64, K] in LDS, the
// transposed weight [K, C] resident in LDS for the workgroup's life, accumulators for the whole dW [C, K] (64 registers per
// lane at 128 x 128) and for the tile's dx [64, K] (32).  Per row tile:
//     park    dy = prologue(dz | dout+argmax, y), x' = act(y_below * scale + shift)  -> LDS       (loads issued a tile ahead)
//     dW     += dy^T x'        both operands by transposed LDS reads (ds_read_b64_tr_b16), as gemm_tn
//     dx      = dy W           A rows from the dy tile, B rows from the resident W^T, as gemm_nt
//     store   dx as bf16 row segments (= dz of the layer below) + ITS BatchNorm-backward sums from (dx, y_below)
// HBM bytes per row: 2C (+2C for a dense dz) + 2K read, 2K written -- against 8C + 6K (dense) for the two-kernel form.
// Persistent over row tiles; every workgroup writes ONE dW slab (summed in slab order by pcb_reduce_slabs, as gemm_tn's) and
// one slab of the sums (pcb_bn_bwd_finalize adds them).  No atomics: reproducible.
constexpr int BF_BM = 64;

// WAVES = 4: C, K <= 128, two workgroups per CU.  WAVES = 8: C <= 256 (K <= 128) -- the last layers of the second
// set-abstraction level -- 131 KB of LDS, one workgroup per CU, the same eight waves; dW tiled 4 x 2 over the waves, dx 2 x 4.
template <int APRO, int TC, int TK, int WAVES>
__global__ __launch_bounds__(WAVES * 64, WAVES == 4 ? 2 : 1) void bwd_fused_kernel(Operand A_arg, Operand X_arg, const u16 *__restrict__ Wt, long R, int C,
                                                           int K, u16 *__restrict__ dx, RedArgs red_arg, float *__restrict__ red_sums,
                                                           float *__restrict__ part)
{
    static_assert(APRO == PRO_DY || APRO == PRO_DY_POOL, "a BatchNorm-backward prologue");
    const Operand A = local_copy(A_arg);
    const Operand X = local_copy(X_arg);
    const float a_slope = act_slope(A.act), x_slope = act_slope(X.act), red_slope = act_slope(red_arg.act);
    constexpr int LDC = TC + 32, LDK = TK + 32;     // row strides of the dy / x' tiles (bf16): conflict-free transposed reads
    constexpr int LDW = TC + 8;                      // row stride of the resident W^T [TK][TC]
    constexpr int LDS_ = TK + 8;                     // dx staging rows (in the x' tile's buffer)
    constexpr int CHC = TC / 8, CHK = TK / 8;        // 16-byte chunks per tile row
    constexpr int THREADS = WAVES * 64;
    constexpr int RPC = THREADS / CHC, RPK = THREADS / CHK;  // rows one pass of the workgroup's threads covers
    constexpr int NC = BF_BM / RPC, NK = BF_BM / RPK;  // chunks per thread and tile
    constexpr int WMB = WAVES / 2, WNB = 2;          // dW [TC, TK] over WMB x WNB waves: (TC / WMB) x (TK / 2) each
    constexpr int XA = TC / WMB / 32, XB = TK / WNB / 32;   // 32-wide MFMA tiles per wave of dW
    constexpr int WNA = WAVES / 2;                   // dx [64, TK] over 2 x WNA waves: 32 rows x (TK / WNA) columns each
    constexpr int XD = TK / WNA / 32;
    static_assert(NC >= 1 && NK >= 1 && XA >= 1 && XB >= 1 && XD >= 1, "tile / wave decomposition");
    static_assert(BF_BM * LDS_ <= BF_BM * LDK, "the dx staging rows live in the x' tile");
    __shared__ __attribute__((aligned(16))) u16 smem[BF_BM * LDC + BF_BM * LDK + TK * LDW];   // 78 KB at 128 x 128: two per CU; 122 KB at 256 x 128
    u16 *const Dy = smem;
    u16 *const Xs = smem + BF_BM * LDC;
    u16 *const Ws = Xs + BF_BM * LDK;
    __shared__ __attribute__((aligned(16))) float cstA[4 * TC];   // scale | shift | p | q of this layer
    __shared__ __attribute__((aligned(16))) float cstX[4 * TK];   // scale | shift | mean | invstd of the layer below

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;       // position in the dW decomposition
    const int wma = wave / WNA, wna = wave % WNA;  // ... in the dx decomposition
    const long tiles = (R + BF_BM - 1) / BF_BM;

    // once per workgroup: constants and W^T (rows = input columns k, contraction index c contiguous; zero outside [K, C])
    for (int i = t; i < TC; i += THREADS) {
        const bool ok = i < C;
        cstA[i] = ok ? A.scale[i] : 0.0f;
        cstA[TC + i] = ok ? A.shift[i] : 0.0f;
        cstA[2 * TC + i] = ok ? A.p[i] : 0.0f;
        cstA[3 * TC + i] = ok ? A.q[i] : 0.0f;
    }
    for (int i = t; i < TK; i += THREADS) {
        const bool ok = i < K;
        cstX[i] = ok ? X.scale[i] : 0.0f;
        cstX[TK + i] = ok ? X.shift[i] : 0.0f;
        cstX[2 * TK + i] = ok ? red_arg.mean[i] : 0.0f;
        cstX[3 * TK + i] = ok ? red_arg.invstd[i] : 0.0f;
    }
    for (int i = t; i < TK * CHC; i += THREADS) {
        const int n = i / CHC, c8 = (i % CHC) * 8;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (n < K && c8 < C) v = *reinterpret_cast<const uint4 *>(Wt + (long)n * C + c8);
        *reinterpret_cast<uint4 *>(&Ws[n * LDW + c8]) = v;
    }

    f32x16 accw[XA][XB];
#pragma unroll
    for (int a = 0; a < XA; ++a)
#pragma unroll
        for (int b = 0; b < XB; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) accw[a][b][i] = 0.0f;

    const int chc = t % CHC, rowc = t / CHC;   // this thread's column chunk / first row of the dy tile
    const int chk = t % CHK, rowk = t / CHK;   // ... of the x' tile (and of the dx rows it stores)
    Raw<APRO> ra[NC];
    Raw<PRO_BNACT> rx[NK];
    auto fetch = [&](long tile) {
        const long m0 = tile * BF_BM;
#pragma unroll
        for (int i = 0; i < NC; ++i) ra[i].load(A, m0 + rowc + RPC * i, chc * 8, R, C);
#pragma unroll
        for (int i = 0; i < NK; ++i) rx[i].load(X, m0 + rowk + RPK * i, chk * 8, R, K);
    };
    const int grp = lane >> 4, gi = lane & 15;
    const int tr_row = 8 * (grp >> 1) + (gi >> 2);
    const int tr_col = 16 * (grp & 1) + 4 * (gi & 3);
    float rs1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, rs2[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // sums of this thread's 8 dx columns

    long tile = blockIdx.x;
    if (tile < tiles) fetch(tile);
    __syncthreads();  // constants, W^T
    // eight-wave form: a wave's W^T fragments of the dx phase do not change from tile to tile -- read once (TC / 16 steps x
    // XD tiles x 4 registers = 64), which halves the LDS traffic of that phase (it is bound by it: 2 KB per MFMA and wave)
    constexpr bool kWreg = false && WAVES == 8 && XD == 1;   // (measured: 64 more registers spill -- 29-51 VGPRs -- at 256 x 128)
    bf16x8 wfrag[kWreg ? TC / 16 : 1];
    if (kWreg) {
#pragma unroll
        for (int ks = 0; ks < TC / 16; ++ks)
            wfrag[ks] = *reinterpret_cast<const bf16x8 *>(&Ws[(wna * (TK / WNA) + (lane & 31)) * LDW + ks * 16 + (lane >> 5) * 8]);
    }
    for (; tile < tiles; tile += gridDim.x) {
        const long m0 = tile * BF_BM;
        {   // park: both tiles, transformed
            ConstsLds<APRO> ka;
            ka.load(cstA, TC, chc * 8);
            ConstsLds<PRO_BNACT> kx;
            kx.load(cstX, TK, chk * 8);
#pragma unroll
            for (int i = 0; i < NC; ++i)
                *reinterpret_cast<uint4 *>(&Dy[(rowc + RPC * i) * LDC + chc * 8]) = finish_with<APRO>(ra[i], ka, a_slope);
#pragma unroll
            for (int i = 0; i < NK; ++i)
                *reinterpret_cast<uint4 *>(&Xs[(rowk + RPK * i) * LDK + chk * 8]) = finish_with<PRO_BNACT>(rx[i], kx, x_slope);
        }
        __syncthreads();
        // eight-wave form (one workgroup per CU: nobody hides a reload's round trip): the raw rows below stay in registers
        // for the sums of the epilogue -- two chunks per thread
        uint4 ykeep[WAVES == 8 ? NK : 1];
        if (WAVES == 8) {
#pragma unroll
            for (int i = 0; i < NK; ++i) ykeep[i] = rx[i].v0;
        }
        if (tile + gridDim.x < tiles) fetch(tile + gridDim.x);   // the next tile's loads fly under everything below
        // dW += dy^T x'   (contraction over the tile's 64 rows)
        typedef __attribute__((address_space(3))) s16x4 *lds_ptr;
#pragma unroll
        for (int ks = 0; ks < BF_BM / 16; ++ks) {
            bf16x8 af[XA], bf[XB];
#pragma unroll
            for (int x = 0; x < XA; ++x) {
                const u16 *pa = &Dy[(ks * 16 + tr_row) * LDC + wm * (TC / WMB) + x * 32 + tr_col];
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)pa);
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(pa + 4 * LDC));
                af[x] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int x = 0; x < XB; ++x) {
                const u16 *pb = &Xs[(ks * 16 + tr_row) * LDK + wn * (TK / 2) + x * 32 + tr_col];
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)pb);
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(pb + 4 * LDK));
                bf[x] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int a = 0; a < XA; ++a)
#pragma unroll
                for (int b = 0; b < XB; ++b)
                    accw[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bf[b], accw[a][b], 0, 0, 0);
        }
        // dx = dy W   (2 x WNA waves: 32 rows x TK / WNA columns each)
        f32x16 accx[XD];
#pragma unroll
        for (int j = 0; j < XD; ++j)
#pragma unroll
            for (int i = 0; i < 16; ++i) accx[j][i] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < TC / 16; ++ks) {
            const int kk = ks * 16 + (lane >> 5) * 8;
            const bf16x8 a = *reinterpret_cast<const bf16x8 *>(&Dy[(wma * 32 + (lane & 31)) * LDC + kk]);
#pragma unroll
            for (int j = 0; j < XD; ++j) {
                const bf16x8 b = kWreg ? wfrag[ks]
                                       : *reinterpret_cast<const bf16x8 *>(&Ws[(wna * (TK / WNA) + j * 32 + (lane & 31)) * LDW + kk]);
                accx[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, accx[j], 0, 0, 0);
            }
        }
        __syncthreads();   // every wave is through with the x' tile: its buffer takes the dx rows
        u16 *const St = Xs;
#pragma unroll
        for (int j = 0; j < XD; ++j)
#pragma unroll
            for (int i = 0; i < 16; ++i)
                St[(wma * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5)) * LDS_ + wna * (TK / WNA) + j * 32 + (lane & 31)] = f2bf(accx[j][i]);
        __syncthreads();
        // store dx (= dz of the layer below) as 16-byte row segments and add its BatchNorm-backward sums: this thread owns
        // the same (row, chunk) positions it loaded y_below at -- the raw rows are requested again (L2: they were read a
        // moment ago) rather than kept in registers across the MFMA phases
        {
            uint4 yraw[NK];
#pragma unroll
            for (int i = 0; i < NK; ++i) {
                const long r = m0 + rowk + RPK * i;
                if (WAVES == 8) yraw[i] = ykeep[i];
                else yraw[i] = *reinterpret_cast<const uint4 *>(X.a0 + (r < R ? r : R - 1) * X.ld + (chk * 8 < K ? chk * 8 : 0));
            }
            float sc[8], sh[8], mu[8], is[8];
            ConstsLds<PRO_BNACT>::rd8(cstX + chk * 8, sc);
            ConstsLds<PRO_BNACT>::rd8(cstX + TK + chk * 8, sh);
            ConstsLds<PRO_BNACT>::rd8(cstX + 2 * TK + chk * 8, mu);
            ConstsLds<PRO_BNACT>::rd8(cstX + 3 * TK + chk * 8, is);
#pragma unroll
            for (int i = 0; i < NK; ++i) {
                const int rr = rowk + RPK * i;
                const long r = m0 + rr;
                const uint4 o = *reinterpret_cast<const uint4 *>(&St[rr * LDS_ + chk * 8]);
                if (r < R && chk * 8 < K) {
                    *reinterpret_cast<uint4 *>(dx + r * K + chk * 8) = o;
                    float dzv[8], yv[8];
                    unpack8(o, dzv);
                    unpack8(yraw[i], yv);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float du = dzv[e] * act_grad(fmaf(yv[e], sc[e], sh[e]), red_slope);
                        rs1[e] += du;
                        rs2[e] = fmaf(du, (yv[e] - mu[e]) * is[e], rs2[e]);
                    }
                }
            }
        }
        __syncthreads();   // the staging rows and the dy tile are rebuilt by the next park
    }

    // the sums of the layer below: threads sharing a column chunk (same t % CHK) meet in LDS, fixed order
    {
        float *const red = reinterpret_cast<float *>(smem);   // [RPK][2][TK] floats over the (idle) dy and x' tiles
        static_assert(RPK * 2 * TK * 4 <= (BF_BM * LDC + BF_BM * LDK) * 2, "the reduction scratch must fit the two tiles");
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            red[(rowk * 2 + 0) * TK + chk * 8 + e] = rs1[e];
            red[(rowk * 2 + 1) * TK + chk * 8 + e] = rs2[e];
        }
        __syncthreads();
        for (int o = t; o < 2 * TK; o += THREADS) {
            const int m = o / TK, c = o % TK;
            float a = 0.0f;
            for (int r = 0; r < RPK; ++r) a += red[(r * 2 + m) * TK + c];
            if (c < K) red_sums[((long)blockIdx.x * 2 + m) * K + c] = a;
        }
    }
    // this workgroup's dW slab (workgroups without a tile write zeros: every slab is summed)
#pragma unroll
    for (int a = 0; a < XA; ++a)
#pragma unroll
        for (int b = 0; b < XB; ++b) {
            const int n = wn * (TK / 2) + b * 32 + (lane & 31);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int m = wm * (TC / WMB) + a * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                if (m < C && n < K) part[(long)blockIdx.x * C * K + (long)m * K + n] = accw[a][b][i];
            }
        }
}

// dy = scale*du + p*y + q written out as rows (the PRO_DY prologue as a pass of its own), for the layers whose
// gradient GEMMs would otherwise rebuild it once per 128-column tile of a WIDE partner matrix: the input-gradient
// GEMM of a layer with K = C > 256 (too wide for the A-resident kernel) and N > 256 inputs walks >= 3 column tiles,
// the weight-gradient GEMM as many tiles again (R = 8192, C = 1024, 1536 inputs: 12 + 12 times).  One lane owns a
// 16-byte column chunk (its constants stay in registers) and walks rows.
__global__ __launch_bounds__(256) void dy_rows_kernel(Operand A_arg, long R, int C, u16 *__restrict__ out)
{
    const Operand A = local_copy(A_arg);
    const float slope = act_slope(A.act);
    const int CT = C / 8, RT = 256 / CT;
    const int cc = threadIdx.x % CT, rl = threadIdx.x / CT;
    if (rl >= RT) return;
    Consts<PRO_DY> k;
    k.load(A, cc * 8, C);
    const long step = (long)gridDim.x * RT;
    for (long r0 = (long)blockIdx.x * RT + rl; r0 < R; r0 += 2 * step) {
        Raw<PRO_DY> ra[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) ra[u].load(A, r0 + u * step, cc * 8, R, C);
#pragma unroll
        for (int u = 0; u < 2; ++u)
            if (r0 + u * step < R) *reinterpret_cast<uint4 *>(out + (r0 + u * step) * C + cc * 8) = ra[u].finish(k, slope);
    }
}

// Gradient of the repeated addends of a STATS == 2 layer (RedArgs.add1 / add2): d add[i, :] = sum of dy over the 2^sh
// consecutive rows coarse row i stood for.  One workgroup per coarse row of the COARSER level (2^sh2 rows, sh2 >= sh1);
// a lane owns a 16-byte column chunk, RT = 256 / (C/8) lanes share the rows of a run, their partial sums meet in LDS.
// dy = bf16(scale.dz.act' + p.y + q), the very values the layer's two gradient GEMMs rebuild in their prologues.
__global__ __launch_bounds__(256) void dy_repeat_sums_kernel(Operand A_arg, long R, int C, int sh1, float *__restrict__ d1,
                                                              int sh2, float *__restrict__ d2)
{
    __shared__ float red[2048];  // [RT][C] = 256 lanes x 8 columns
    const Operand A = local_copy(A_arg);
    const float slope = act_slope(A.act);
    const int CT = C / 8, RT = 256 / CT;
    const int cc = threadIdx.x % CT, rl = threadIdx.x / CT;
    const bool on = rl < RT;
    Consts<PRO_DY> k;
    k.load(A, cc * 8, C);
    const long g1 = 1L << sh1, g2 = 1L << sh2;
    const long base = (long)blockIdx.x * g2;
    float tot = 0.0f;  // lanes t < C: column t of the coarser level's sum
    for (long s = 0; s < g2; s += g1) {
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (on) {
            for (long r0 = rl; r0 < g1; r0 += 2 * RT) {
                Raw<PRO_DY> ra[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) ra[u].load(A, base + s + r0 + u * RT, cc * 8, R, C);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    if (r0 + u * RT >= g1 || base + s + r0 + u * RT >= R) continue;
                    float f[8];
                    unpack8(ra[u].finish(k, slope), f);
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[i] += f[i];
                }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) red[rl * C + cc * 8 + i] = acc[i];
        }
        __syncthreads();
        if ((int)threadIdx.x < C) {
            float v = 0.0f;
            for (int j = 0; j < RT; ++j) v += red[j * C + threadIdx.x];
            if (base + s < R) d1[((base + s) >> sh1) * C + threadIdx.x] = v;
            tot += v;
        }
        __syncthreads();
    }
    if (d2 && (int)threadIdx.x < C && base < R) d2[(base >> sh2) * C + threadIdx.x] = tot;
}

template <int PRO>
void launch_nt(const Operand &A, const u16 *Bw, long R, int N, int K, u16 *out, float *sums, int nparts, hipStream_t st,
               const RedArgs *red = nullptr, const float *centre = nullptr)
{
    // with slabs the caller's count IS the grid (it sized its buffer and its finalize call for it);
    // without, the resident-workgroup preference for the hint in force now
    const unsigned ny = (unsigned)((N + NT_BN - 1) / NT_BN);
    const dim3 grid((unsigned)(sums ? nparts : pcb_nt_grid_x(PRO, R, N, pcb_busy_cus())), ny);
    RedArgs none = {nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr};
    none.centre = sums ? centre : nullptr;
    // the eight-wave form: its LDS constants limit K for the BatchNorm-on-load prologue only (plain rows: any K)
    if (PRO == PRO_PLAIN || (PRO == PRO_BNACT && K <= N8_MAXK)) {
        // forward prologues: the eight-wave form, two workgroups per CU
        if (sums)
            hipLaunchKernelGGL((gemm_nt8_kernel<PRO, 1, 0, 4, 4>), grid, dim3(512), 0, st, A, Bw, R, N, K, out, sums, none);
        else
            hipLaunchKernelGGL((gemm_nt8_kernel<PRO, 0, 0, 4, 4>), grid, dim3(512), 0, st, A, Bw, R, N, K, out, sums, none);
        return;
    }
    if (red && PRO >= PRO_DY)
        hipLaunchKernelGGL((gemm_nt_kernel<PRO, 0, (PRO >= PRO_DY)>), grid, dim3(256), 0, st, A, Bw, R, N, K, out, sums, *red);
    else if (sums)
        hipLaunchKernelGGL((gemm_nt_kernel<PRO, 1, 0>), grid, dim3(256), 0, st, A, Bw, R, N, K, out, sums, none);
    else
        hipLaunchKernelGGL((gemm_nt_kernel<PRO, 0, 0>), grid, dim3(256), 0, st, A, Bw, R, N, K, out, sums, none);
}

template <int APRO>
void launch_tn(const Operand &A, const Operand &B, int bpro, long R, int M, int N, float *part, float *dW,
               int out_cols, int out_perm, hipStream_t st, float *colsum = nullptr)
{
    long rps;
    static const bool trace = getenv("PCB_TN_TRACE") != nullptr;  // launch list for tools/tn_bench.py
    if (trace) fprintf(stderr, "[pcb_tn] %d %d %ld %d %d %d\n", APRO, bpro, R, M, N, colsum ? 1 : 0);
    // fewer splits while another kernel holds CUs (never more than pcb_gemm_tn_workspace assumed)
    // two four-wave teams per workgroup (one slab for both) unless PCB_TN_TEAMS=1 asks for the one-team form: half as
    // many workgroups of twice the size, the same waves per CU (never more splits than pcb_gemm_tn_workspace assumed)
    static const bool one_team_env = getenv("PCB_TN_TEAMS") && atoi(getenv("PCB_TN_TEAMS")) == 1;
    const long target = 512 - 2 * pcb_busy_cus();
    static const bool no_narrow = getenv("PCB_TN_NARROW") && atoi(getenv("PCB_TN_NARROW")) == 0;
    // an operand of at most 64 columns gets a 64-wide tile (one tile either way: the split count is unchanged)
    const bool na = !one_team_env && !no_narrow && M <= 64, nb = !one_team_env && !no_narrow && N <= 64;
    const int tm = na ? 1 : (M + TN_BM - 1) / TN_BM, tn = nb ? 1 : (N + TN_BN - 1) / TN_BN;
    // a two-team workgroup (80 KB of LDS) has a CU to itself: the grid must not exceed the CUs it may use, or its last
    // few workgroups run as a second round (258 workgroups for 6 tiles x 43 splits took 1.45x the time of 252); matrices
    // of more than 64 tiles would leave a quarter of the chip idle that way and keep the one-team form
    const bool one_team = one_team_env || tm * tn > 64;
    const long splits = pcb_tn_splits(R, M, N, &rps, one_team ? target : ((target / 2) / (tm * tn) > 0 ? (target / 2) / (tm * tn) : 1) * (tm * tn));
    const dim3 grid((unsigned)(tm * tn * splits));
    const long stride = (long)M * N + (colsum ? M : 0);
    if (one_team) {
        if (colsum && APRO == PRO_PLAIN && bpro == PRO_PLAIN)
            hipLaunchKernelGGL((gemm_tn_kernel<PRO_PLAIN, PRO_PLAIN, 1>), grid, dim3(256), 0, st, A, B, R, M, N, rps, part, stride, tm, tn);
        else if (bpro == PRO_PLAIN)
            hipLaunchKernelGGL((gemm_tn_kernel<APRO, PRO_PLAIN>), grid, dim3(256), 0, st, A, B, R, M, N, rps, part, stride, tm, tn);
        else
            hipLaunchKernelGGL((gemm_tn_kernel<APRO, PRO_BNACT>), grid, dim3(256), 0, st, A, B, R, M, N, rps, part, stride, tm, tn);
    } else if (na || nb) {
#define PCB_TN_LAUNCH(TA, TB)                                                                                                   \
    do {                                                                                                                        \
        if (colsum && APRO == PRO_PLAIN && bpro == PRO_PLAIN)                                                                   \
            hipLaunchKernelGGL((gemm_tn_kernel<PRO_PLAIN, PRO_PLAIN, 1, 2, TA, TB>), grid, dim3(512), 0, st, A, B, R, M, N, rps, \
                               part, stride, tm, tn);                                                                           \
        else if (bpro == PRO_PLAIN)                                                                                             \
            hipLaunchKernelGGL((gemm_tn_kernel<APRO, PRO_PLAIN, 0, 2, TA, TB>), grid, dim3(512), 0, st, A, B, R, M, N, rps, part, \
                               stride, tm, tn);                                                                                 \
        else                                                                                                                    \
            hipLaunchKernelGGL((gemm_tn_kernel<APRO, PRO_BNACT, 0, 2, TA, TB>), grid, dim3(512), 0, st, A, B, R, M, N, rps, part, \
                               stride, tm, tn);                                                                                 \
    } while (0)
        if (na && nb)
            PCB_TN_LAUNCH(64, 64);
        else if (na)
            PCB_TN_LAUNCH(64, 128);
        else
            PCB_TN_LAUNCH(128, 64);
#undef PCB_TN_LAUNCH
    } else if (colsum && APRO == PRO_PLAIN && bpro == PRO_PLAIN)
        hipLaunchKernelGGL((gemm_tn_kernel<PRO_PLAIN, PRO_PLAIN, 1, 2>), grid, dim3(512), 0, st, A, B, R, M, N, rps, part, stride, tm, tn);
    else if (bpro == PRO_PLAIN)
        hipLaunchKernelGGL((gemm_tn_kernel<APRO, PRO_PLAIN, 0, 2>), grid, dim3(512), 0, st, A, B, R, M, N, rps, part, stride, tm, tn);
    else
        hipLaunchKernelGGL((gemm_tn_kernel<APRO, PRO_BNACT, 0, 2>), grid, dim3(512), 0, st, A, B, R, M, N, rps, part, stride, tm, tn);
    (void)pcb_reduce_slabs_vec(part, (int)splits, stride, (long)M * N, dW, N, out_cols, out_perm, 8, colsum, M, st);
}

inline bool bad_dim(long v) { return v <= 0 || (v & 7) != 0; }

// Algorithmic HBM bytes of one gemm_nt launch: the A operand as its prologue reads it + the bf16
// output (the weights, a few KB, are not counted).  This is the figure bench.py's roofline uses.
inline double nt_bytes(int pro, long R, int N, int K, int ns)
{
    double a = 2.0 * R * K;
    if (pro == PRO_DY) a = 4.0 * R * K;
    if (pro == PRO_DY_POOL) a = 2.0 * R * K + 5.0 * (double)(R / (ns > 0 ? ns : 1)) * K;
    return a + 2.0 * R * N;
}

}  // namespace

// A-operand description shared by the C entry points (all pointers may be NULL where unused):
//   pro 0 plain: a0;  1 BN+act on load: a0, scale, shift, act;
//   2 dy from dense dz: a0 = dz, a1 = y, scale, shift, p, q, act;
//   3 dy from a pooled layer: dout, arg, ns, a1 = y, scale, shift, p, q, act.
static Operand make_operand(const void *a0, const void *a1, long ld, const float *scale, const float *shift,
                            const float *p, const float *q, const float *dout, const unsigned char *arg,
                            int ns, int act)
{
    Operand o;
    o.a0 = (const u16 *)a0;
    o.a1 = (const u16 *)a1;
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

extern "C" int pcb_gemm_nt_bf16(int pro, const void *a0, const void *a1, const float *scale,
                                const float *shift, const float *p, const float *q, const float *dout,
                                const unsigned char *argmax, int ns, int act, const void *w, long R, int N,
                                int K, void *out, float *sums, int nparts, void *stream)
{
    if (!w || !out || R <= 0) return PCB_ERR_INVALID_ARG;
    if (sums && (nparts < 1 || nparts > PCB_MAX_SLABS)) return PCB_ERR_INVALID_ARG;
    if (bad_dim(N) || bad_dim(K)) return PCB_ERR_UNSUPPORTED;
    if (pro < 0 || pro > 3) return PCB_ERR_INVALID_ARG;
    if ((pro <= PRO_DY && !a0) || (pro >= PRO_DY && !a1) || (pro >= PRO_BNACT && (!scale || !shift))) return PCB_ERR_INVALID_ARG;
    if (pro >= PRO_DY && (!p || !q)) return PCB_ERR_INVALID_ARG;
    if (pro == PRO_DY_POOL && (!dout || !argmax || ns <= 0 || ns > 255)) return PCB_ERR_INVALID_ARG;
    const Operand A = make_operand(a0, a1, K, scale, shift, p, q, dout, argmax, ns, act);
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t timed;
    pcb_timer_begin(st, &timed);
    const double bytes = nt_bytes(pro, R, N, K, ns);
    if (pro >= PRO_DY && !sums && N > NT_BN && K <= 256) {
        // wide input gradient: transformed A tile resident in LDS, column tiles walked inside
        const long tiles = (R + AR_BM - 1) / AR_BM;
        const long ares_chip = pcb_nt_grid_x(PRO_DY, (long)PCB_MAX_SLABS * NT_BM, NT_BN, pcb_busy_cus());  // the backward grid of a full chip
        static const long ares_wgs = getenv("PCB_ARES_WGS") ? atol(getenv("PCB_ARES_WGS")) : 3;  // resident workgroups per CU
        const long ares_grid = ares_chip * ares_wgs / 2;   // (ares_chip counts two per CU)
        const dim3 grid((unsigned)(tiles < ares_grid ? tiles : ares_grid));
        if (pro == PRO_DY) {
            if (K <= 128)
                hipLaunchKernelGGL((gemm_nt_ares_kernel<PRO_DY, 16>), grid, dim3(256), 0, st, A, (const u16 *)w, R, N, K, (u16 *)out);
            else
                hipLaunchKernelGGL((gemm_nt_ares_kernel<PRO_DY, 32>), grid, dim3(256), 0, st, A, (const u16 *)w, R, N, K, (u16 *)out);
        } else {
            if (K <= 128)
                hipLaunchKernelGGL((gemm_nt_ares_kernel<PRO_DY_POOL, 16>), grid, dim3(256), 0, st, A, (const u16 *)w, R, N, K, (u16 *)out);
            else
                hipLaunchKernelGGL((gemm_nt_ares_kernel<PRO_DY_POOL, 32>), grid, dim3(256), 0, st, A, (const u16 *)w, R, N, K, (u16 *)out);
        }
        pcb_timer_end(st, timed, bytes, pro, R, N, K);
        return pcb_check_launch();
    }
    switch (pro) {
        case PRO_PLAIN: launch_nt<PRO_PLAIN>(A, (const u16 *)w, R, N, K, (u16 *)out, sums, nparts, st); break;
        case PRO_BNACT: launch_nt<PRO_BNACT>(A, (const u16 *)w, R, N, K, (u16 *)out, sums, nparts, st); break;
        case PRO_DY: launch_nt<PRO_DY>(A, (const u16 *)w, R, N, K, (u16 *)out, sums, nparts, st); break;
        default: launch_nt<PRO_DY_POOL>(A, (const u16 *)w, R, N, K, (u16 *)out, sums, nparts, st); break;
    }
    pcb_timer_end(st, timed, bytes, pro, R, N, K);
    return pcb_check_launch();
}

// The forward GEMM of a Conv + BatchNorm layer with its rows stored CENTRED: out = bf16(A'.W^T - centre[N]) and the
// statistics slabs of those rows (pro 0 plain / 1 BatchNorm+activation of the previous layer on load).  See RedArgs.
extern "C" int pcb_gemm_nt_stats_bf16(int pro, const void *a, const float *scale, const float *shift, int act,
                                      const void *w, long R, int N, int K, void *out, float *sums, int nparts,
                                      const float *centre, void *stream)
{
    if (!a || !w || !out || !sums || R <= 0) return PCB_ERR_INVALID_ARG;
    if (nparts < 1 || nparts > PCB_MAX_SLABS) return PCB_ERR_INVALID_ARG;
    if (bad_dim(N) || bad_dim(K)) return PCB_ERR_UNSUPPORTED;
    if (pro != PRO_PLAIN && pro != PRO_BNACT) return PCB_ERR_INVALID_ARG;
    if (pro == PRO_BNACT && (!scale || !shift)) return PCB_ERR_INVALID_ARG;
    const Operand A = make_operand(a, nullptr, K, scale, shift, nullptr, nullptr, nullptr, nullptr, 1, act);
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t timed;
    pcb_timer_begin(st, &timed);
    if (pro == PRO_PLAIN)
        launch_nt<PRO_PLAIN>(A, (const u16 *)w, R, N, K, (u16 *)out, sums, nparts, st, nullptr, centre);
    else
        launch_nt<PRO_BNACT>(A, (const u16 *)w, R, N, K, (u16 *)out, sums, nparts, st, nullptr, centre);
    pcb_timer_end(st, timed, nt_bytes(pro, R, N, K, 1), pro, R, N, K);
    return pcb_check_launch();
}

extern "C" int pcb_dy_rows_bf16(const void *dz, const void *y, const float *scale, const float *shift, const float *p,
                                const float *q, int act, long R, int C, void *dy, void *stream)
{
    if (!dz || !y || !scale || !shift || !p || !q || !dy || R <= 0) return PCB_ERR_INVALID_ARG;
    if (bad_dim(C) || C > 2048) return PCB_ERR_UNSUPPORTED;
    const Operand A = make_operand(dz, y, C, scale, shift, p, q, nullptr, nullptr, 1, act);
    const long RT = 256 / (C / 8);
    long blocks = (R + RT * 4 - 1) / (RT * 4);
    blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
    hipLaunchKernelGGL(dy_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, A, R, C, (u16 *)dy);
    pcb_account(6.0 * R * C);
    return pcb_check_launch();
}

extern "C" int pcb_gemm_nt_f32out_bf16(const void *a, const void *w, long R, int N, int K, float *out, void *stream)
{
    if (!a || !w || !out || R <= 0) return PCB_ERR_INVALID_ARG;
    if (bad_dim(N) || bad_dim(K)) return PCB_ERR_UNSUPPORTED;
    const Operand A = make_operand(a, nullptr, K, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1, 0);
    const unsigned ny = (unsigned)((N + NT_BN - 1) / NT_BN);
    const dim3 grid((unsigned)pcb_nt_grid_x(PRO_PLAIN, R, N, pcb_busy_cus()), ny);
    const RedArgs none = {nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr};
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t timed;
    pcb_timer_begin(st, &timed);
    hipLaunchKernelGGL((gemm_nt_kernel<PRO_PLAIN, 0, 0, 1>), grid, dim3(256), 0, st, A, (const u16 *)w, R, N, K, (u16 *)out,
                       (float *)nullptr, none);
    pcb_timer_end(st, timed, 2.0 * R * K + 4.0 * R * N, 20, R, N, K);
    return pcb_check_launch();
}

// y = x W^T + b for a conv without BatchNorm: the plain GEMM with the bias added to the fp32
// accumulators before rounding.
extern "C" int pcb_gemm_nt_bias_bf16(const void *a, const void *w, const float *bias, long R, int N, int K, void *out,
                                     void *stream)
{
    if (!a || !w || !out || R <= 0) return PCB_ERR_INVALID_ARG;
    if (bad_dim(N) || bad_dim(K)) return PCB_ERR_UNSUPPORTED;
    const Operand A = make_operand(a, nullptr, K, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1, 0);
    const unsigned ny = (unsigned)((N + NT_BN - 1) / NT_BN);
    const dim3 grid((unsigned)pcb_nt_grid_x(PRO_PLAIN, R, N, pcb_busy_cus()), ny);
    const RedArgs epi = {nullptr, nullptr, nullptr, nullptr, nullptr, 0, bias, nullptr};
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t timed;
    pcb_timer_begin(st, &timed);
    hipLaunchKernelGGL((gemm_nt8_kernel<PRO_PLAIN, 0, 0, 4, 4>), grid, dim3(512), 0, st, A, (const u16 *)w, R, N, K,
                       (u16 *)out, (float *)nullptr, epi);
    pcb_timer_end(st, timed, nt_bytes(PRO_PLAIN, R, N, K, 0), PRO_PLAIN, R, N, K);
    return pcb_check_launch();
}

extern "C" int pcb_gemm_tn_bf16(int apro, const void *dz, const void *y, const float *scale,
                                const float *shift, const float *p, const float *q, const float *dout,
                                const unsigned char *argmax, int ns, int act, int bpro, const void *x,
                                const float *xscale, const float *xshift, int xact, long R, int M, int N,
                                float *workspace, float *dW, int out_cols, int out_perm, void *stream)
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
    const Operand A = make_operand(dz, y, M, scale, shift, p, q, dout, argmax, ns, act);
    const Operand B = make_operand(x, nullptr, N, xscale, xshift, nullptr, nullptr, nullptr, nullptr, 1, xact);
    hipStream_t st = (hipStream_t)stream;
    if (apro == PRO_PLAIN)
        launch_tn<PRO_PLAIN>(A, B, bpro, R, M, N, workspace, dW, out_cols, out_perm, st);
    else if (apro == PRO_DY)
        launch_tn<PRO_DY>(A, B, bpro, R, M, N, workspace, dW, out_cols, out_perm, st);
    else
        launch_tn<PRO_DY_POOL>(A, B, bpro, R, M, N, workspace, dW, out_cols, out_perm, st);
    pcb_account((apro == PRO_DY ? 4.0 : 2.0) * R * M + (apro == PRO_DY_POOL ? 5.0 * (double)(R / (ns > 0 ? ns : 1)) * M : 0.0) + 2.0 * R * N);
    return pcb_check_launch();
}

// y = (x W^T + b) + res: the same, and the bf16 rows `res` [R, N] are added to the rounded result in the epilogue -- the
// sum of two branches' outputs (EnhancedFeaturePropagation: trunk + boundary term, pointnet2_utils.py:296) without a
// pass of its own.
extern "C" int pcb_gemm_nt_bias_add_bf16(const void *a, const void *w, const float *bias, const void *res, long R, int N,
                                         int K, void *out, void *stream)
{
    if (!a || !w || !out || !res || R <= 0) return PCB_ERR_INVALID_ARG;
    if (bad_dim(N) || bad_dim(K)) return PCB_ERR_UNSUPPORTED;
    const Operand A = make_operand(a, nullptr, K, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1, 0);
    const unsigned ny = (unsigned)((N + NT_BN - 1) / NT_BN);
    const dim3 grid((unsigned)pcb_nt_grid_x(PRO_PLAIN, R, N, pcb_busy_cus()), ny);
    const RedArgs epi = {nullptr, nullptr, nullptr, nullptr, nullptr, 0, bias, (const u16 *)res};
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t timed;
    pcb_timer_begin(st, &timed);
    hipLaunchKernelGGL((gemm_nt8_kernel<PRO_PLAIN, 0, 0, 4, 4>), grid, dim3(512), 0, st, A, (const u16 *)w, R, N, K,
                       (u16 *)out, (float *)nullptr, epi);
    pcb_timer_end(st, timed, nt_bytes(PRO_PLAIN, R, N, K, 0) + 2.0 * R * N, PRO_PLAIN, R, N, K);
    return pcb_check_launch();
}

// The GEMM of a BatchNorm layer whose input is [x | level1 repeated 2^sh1 times | level2 repeated 2^sh2 times] with only x
// as the A operand: out = bf16(x W^T + add1[r >> sh1] + add2[r >> sh2]) and the column sums / sums of squares of out
// as slabs (one per workgroup along x, nparts of them), see RedArgs.  add2 may be NULL.
extern "C" int pcb_gemm_nt_stats_add_bf16(const void *a, const void *w, long R, int N, int K, void *out, float *sums,
                                          int nparts, const float *add1, int sh1, const float *add2, int sh2,
                                          const float *centre, void *stream)
{
    if (!a || !w || !out || !sums || !add1 || R <= 0) return PCB_ERR_INVALID_ARG;
    if (nparts < 1 || nparts > PCB_MAX_SLABS) return PCB_ERR_INVALID_ARG;
    if (bad_dim(N) || bad_dim(K)) return PCB_ERR_UNSUPPORTED;
    if (sh1 < 2 || sh1 > 30 || (add2 && (sh2 < 2 || sh2 > 30))) return PCB_ERR_UNSUPPORTED;
    if ((R & ((1L << sh1) - 1)) || (add2 && (R & ((1L << sh2) - 1)))) return PCB_ERR_INVALID_ARG;
    const Operand A = make_operand(a, nullptr, K, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1, 0);
    const unsigned ny = (unsigned)((N + NT_BN - 1) / NT_BN);
    const dim3 grid((unsigned)nparts, ny);
    const RedArgs epi = {nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, add1, add2, sh1, add2 ? sh2 : sh1, centre};
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t timed;
    pcb_timer_begin(st, &timed);
    hipLaunchKernelGGL((gemm_nt8_kernel<PRO_PLAIN, 2, 0, 4, 4>), grid, dim3(512), 0, st, A, (const u16 *)w, R, N, K,
                       (u16 *)out, sums, epi);
    pcb_timer_end(st, timed, nt_bytes(PRO_PLAIN, R, N, K, 0) + 4.0 * (double)(R >> sh1) * N + (add2 ? 4.0 * (double)(R >> sh2) * N : 0.0),
                  PRO_PLAIN, R, N, K);
    return pcb_check_launch();
}

// d add1 [R >> sh1, C], d add2 [R >> sh2, C] (fp32; d2 may be NULL, else sh2 >= sh1) of such a layer from its dz, y and the
// constants of its BatchNorm backward (the arguments of pcb_dy_rows_bf16).
extern "C" int pcb_dy_repeat_sums_bf16(const void *dz, const void *y, const float *scale, const float *shift, const float *p,
                                       const float *q, int act, long R, int C, int sh1, float *d1, int sh2, float *d2,
                                       void *stream)
{
    if (!dz || !y || !scale || !shift || !p || !q || !d1 || R <= 0) return PCB_ERR_INVALID_ARG;
    if (bad_dim(C) || C > 2048) return PCB_ERR_UNSUPPORTED;
    if (!d2) sh2 = sh1;
    if (sh1 < 0 || sh2 < sh1 || sh2 > 30) return PCB_ERR_INVALID_ARG;
    if (R & ((1L << sh2) - 1)) return PCB_ERR_INVALID_ARG;
    const Operand A = make_operand(dz, y, C, scale, shift, p, q, nullptr, nullptr, 1, act);
    hipLaunchKernelGGL(dy_repeat_sums_kernel, dim3((unsigned)(R >> sh2)), dim3(256), 0, (hipStream_t)stream, A, R, C, sh1,
                       d1, sh2, d2);
    pcb_account(4.0 * R * C + 4.0 * (double)(R >> sh1) * C + (d2 ? 4.0 * (double)(R >> sh2) * C : 0.0));
    return pcb_check_launch();
}

// Weight AND bias gradient of a conv without BatchNorm in one pass over dy: dW = dy^T x, dbias = column sums of dy.
extern "C" int pcb_gemm_tn_bias_bf16(const void *dy, const void *x, long R, int M, int N, float *workspace, float *dW,
                                     int out_cols, int out_perm, float *dbias, void *stream)
{
    if (!dy || !x || !dW || !dbias || !workspace || R <= 0) return PCB_ERR_INVALID_ARG;
    if (out_cols <= 0) { out_cols = N; out_perm = 0; }
    if (out_cols > N) return PCB_ERR_INVALID_ARG;
    if (bad_dim(M) || bad_dim(N)) return PCB_ERR_UNSUPPORTED;
    const Operand A = make_operand(dy, nullptr, M, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1, 0);
    const Operand B = make_operand(x, nullptr, N, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1, 0);
    launch_tn<PRO_PLAIN>(A, B, PRO_PLAIN, R, M, N, workspace, dW, out_cols, out_perm, (hipStream_t)stream, dbias);
    pcb_account(2.0 * R * M + 2.0 * R * N);
    return pcb_check_launch();
}

// One layer's backward in one pass (bwd_fused_kernel): dx [R,K] = dy . W (bf16, = dz of the layer below), dW = dy^T . x' as
// `nparts` slabs [C*K] in `workspace` summed into dW [C, out_cols] (weight layout, as pcb_gemm_tn_bf16), and the layer
// below's BatchNorm-backward sums as `nparts` slabs red_sums [nparts][2][K].  dy from (dz | dout + argmax, y) with
// scale / shift / p / q (pro 2 / 3 as pcb_gemm_nt_bf16); x' = act(x * xscale + xshift) with the layer below's raw rows x
// (= red_y: the same tensor) and its mean / invstd.  wt = the layer's prepared transposed weight [K, C].
// C, K multiples of 8, C <= 256, K <= 128; 1 <= nparts <= PCB_MAX_SLABS (the grid).
extern "C" int pcb_bwd_fused_supported(int C, int K) { return C > 0 && K > 0 && !(C & 7) && !(K & 7) && C <= 256 && K <= 128; }

extern "C" int pcb_bwd_fused_bf16(int pro, const void *dz, const void *y, const float *scale, const float *shift, const float *p,
                                  const float *q, const float *dout, const unsigned char *argmax, int ns, int act,
                                  const void *wt, const void *x, const float *xscale, const float *xshift,
                                  const float *xmean, const float *xinvstd, int xact, long R, int C, int K, void *dx,
                                  float *red_sums, int nparts, float *workspace, float *dW, int out_cols, int out_perm,
                                  void *stream)
{
    if (!y || !scale || !shift || !p || !q || !wt || !x || !xscale || !xshift || !xmean || !xinvstd || !dx || !red_sums ||
        !workspace || !dW || R <= 0)
        return PCB_ERR_INVALID_ARG;
    if (pro != PRO_DY && pro != PRO_DY_POOL) return PCB_ERR_INVALID_ARG;
    if (pro == PRO_DY ? !dz : (!dout || !argmax || ns <= 0 || ns > 255)) return PCB_ERR_INVALID_ARG;
    if (!pcb_bwd_fused_supported(C, K)) return PCB_ERR_UNSUPPORTED;
    if (nparts < 1 || nparts > PCB_MAX_SLABS) return PCB_ERR_INVALID_ARG;
    if (out_cols <= 0) { out_cols = K; out_perm = 0; }
    if (out_cols > K) return PCB_ERR_INVALID_ARG;
    const Operand A = make_operand(dz, y, C, scale, shift, p, q, dout, argmax, ns, act);
    const Operand X = make_operand(x, nullptr, K, xscale, xshift, nullptr, nullptr, nullptr, nullptr, 1, xact);
    RedArgs red = {(const u16 *)x, xscale, xshift, xmean, xinvstd, xact, nullptr, nullptr};
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t timed;
    pcb_timer_begin(st, &timed);
    const dim3 grid((unsigned)nparts);
#define PCB_BF_LAUNCH(P, TC_, TK_, W_)                                                                                              \
    hipLaunchKernelGGL((bwd_fused_kernel<P, TC_, TK_, W_>), grid, dim3(W_ * 64), 0, st, A, X, (const u16 *)wt, R, C, K, (u16 *)dx, \
                       red, red_sums, workspace)
    const bool wc = C > 64, wk = K > 64;
    if (pro == PRO_DY) {
        if (C > 128) PCB_BF_LAUNCH(PRO_DY, 256, 128, 8);
        else if (wc && wk) PCB_BF_LAUNCH(PRO_DY, 128, 128, 4);
        else if (wc) PCB_BF_LAUNCH(PRO_DY, 128, 64, 4);
        else if (wk) PCB_BF_LAUNCH(PRO_DY, 64, 128, 4);
        else PCB_BF_LAUNCH(PRO_DY, 64, 64, 4);
    } else {
        if (C > 128) PCB_BF_LAUNCH(PRO_DY_POOL, 256, 128, 8);
        else if (wc && wk) PCB_BF_LAUNCH(PRO_DY_POOL, 128, 128, 4);
        else if (wc) PCB_BF_LAUNCH(PRO_DY_POOL, 128, 64, 4);
        else if (wk) PCB_BF_LAUNCH(PRO_DY_POOL, 64, 128, 4);
        else PCB_BF_LAUNCH(PRO_DY_POOL, 64, 64, 4);
    }
#undef PCB_BF_LAUNCH
    // algorithmic bytes: the dy operand as its prologue reads it, the rows of the layer below once, dx written once
    const double bytes = nt_bytes(pro, R, 0, C, ns) + 2.0 * R * K + 2.0 * R * K;
    pcb_timer_end(st, timed, bytes, pro + 20, R, K, C);
    const int status = pcb_check_launch();
    if (status != PCB_OK) return status;
    return pcb_reduce_slabs(workspace, nparts, (long)C * K, dW, K, out_cols, out_perm, 8, st);
}

// pcb_gemm_nt_bf16 for an input-gradient GEMM (pro 2 or 3) that ALSO accumulates the BatchNorm
// backward sums of the layer below (see RedArgs): red_sums is [nparts][2][N], one slab per workgroup along x.
extern "C" int pcb_gemm_nt_red_bf16(int pro, const void *a0, const void *a1, const float *scale,
                                    const float *shift, const float *p, const float *q, const float *dout,
                                    const unsigned char *argmax, int ns, int act, const void *w, long R, int N,
                                    int K, void *out, const void *red_y, const float *red_scale,
                                    const float *red_shift, const float *red_mean, const float *red_invstd,
                                    int red_act, float *red_sums, int nparts, void *stream)
{
    if (!w || !out || R <= 0 || !red_y || !red_scale || !red_shift || !red_mean || !red_invstd || !red_sums)
        return PCB_ERR_INVALID_ARG;
    if (nparts < 1 || nparts > PCB_MAX_SLABS) return PCB_ERR_INVALID_ARG;
    if (bad_dim(N) || bad_dim(K)) return PCB_ERR_UNSUPPORTED;
    if (pro != PRO_DY && pro != PRO_DY_POOL) return PCB_ERR_INVALID_ARG;
    if (!a1 || !scale || !shift || !p || !q || (pro == PRO_DY && !a0)) return PCB_ERR_INVALID_ARG;
    if (pro == PRO_DY_POOL && (!dout || !argmax || ns <= 0 || ns > 255)) return PCB_ERR_INVALID_ARG;
    if (N > NT_BN) return PCB_ERR_UNSUPPORTED;  // one column tile only (the caller checks N <= 128)
    const Operand A = make_operand(a0, a1, K, scale, shift, p, q, dout, argmax, ns, act);
    const RedArgs red = {(const u16 *)red_y, red_scale, red_shift, red_mean, red_invstd, red_act, nullptr, nullptr};
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t timed;
    pcb_timer_begin(st, &timed);
    if (pro == PRO_DY)
        launch_nt<PRO_DY>(A, (const u16 *)w, R, N, K, (u16 *)out, red_sums, nparts, st, &red);
    else
        launch_nt<PRO_DY_POOL>(A, (const u16 *)w, R, N, K, (u16 *)out, red_sums, nparts, st, &red);
    pcb_timer_end(st, timed, nt_bytes(pro, R, N, K, ns) + 2.0 * R * N, pro + 10, R, N, K);
    return pcb_check_launch();
}
