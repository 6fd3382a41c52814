// Global scaled-dot-product attention for gfx950 -- row f4 of the scope table.
//
// Replaces F.scaled_dot_product_attention(q, k, v) in PointAttention.forward,
// models/PointTransformerV3.py:64-117 (the call at :102; inference_ptv3.py:101-105 builds the network with
// embed 384, 2 heads => head_dim 192): every point attends to every point of its scene.  The reference leaves
// the kernel to ATen; here it is one flash-attention pass, forward only (cfg5 is an inference configuration):
// nothing of size N x N is ever stored.
//
// Work split: one wave owns 32 queries, ONE PER LANE (both lane halves hold the same 32), for all keys.
//   S^T tile [32 keys x 32 queries] = K . Q^T on v_mfma_f32_32x32x16_bf16: A = K rows from LDS, B = Q^T held in
//   registers for the whole kernel (loaded straight from global memory: a lane's query row, 16 bytes per k-step).
//   The accumulator layout puts the QUERY on the lane and the 16 keys of a half on the registers, so the
//   running max / sum of the online softmax are per-lane scalars: one max over 16 registers plus ONE exchange
//   with the other lane half, no cross-lane reductions, and the rescaling of O by exp(m_old - m_new) is a
//   per-lane scalar too.
//   O^T [D x 32 queries] += V^T . P^T: B = P^T is the S^T accumulator itself, rounded to bf16 -- the register
//   order of a 32x32 accumulator (rows 4h + (j&3) + 8(j>>2) for element j of half h) is used as the k-order of
//   the product, so no lane movement; A = V^T comes out of the row-major V tile in LDS with the hardware
//   transpose read (ds_read_b64_tr_b16) at the matching row blocks (4h.., 4h+8..).
// A workgroup = 8 waves = 256 queries of one (scene, head) shares the K / V tiles of 64 keys, double-buffered
// in LDS, the next tile's rows in flight in registers during the current tile's MFMAs; one barrier per tile.
// Input is the qkv projection as the reference produces it, [B, N, 3, H, D] (models/PointTransformerV3.py:96),
// output [B, N, H*D] = the layout after the reference's transpose + reshape (:113): no permute copies.
#include "pcb_common.h"

namespace {

typedef unsigned short u16;
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int AT_WAVES = 8;
constexpr int AT_THREADS = AT_WAVES * 64;
constexpr int AT_QW = 32;                    // queries per wave
constexpr int AT_QB = AT_WAVES * AT_QW;      // queries per workgroup
constexpr int AT_KT = 64;                    // keys per tile

__device__ __forceinline__ u16 f2bf(float f) { return __builtin_bit_cast(u16, (__bf16)f); }

template <int D>
__global__ __launch_bounds__(AT_THREADS, 1) void attention_fwd_bf16_kernel(const u16 *__restrict__ qkv, int N, int H,
                                                                          float scale_log2e, u16 *__restrict__ out)
{
    static_assert(D % 32 == 0 && D <= 256, "head dimension: a multiple of 32 up to 256");
    constexpr int KS = D / 16;        // k-steps of Q.K^T
    constexpr int DT = D / 32;        // 32-row tiles of O^T
    constexpr int CH = D / 8;         // 16-byte chunks per row
    constexpr int KLD = D + 8;        // K tile row stride (bf16): (D/8 + 1) odd -> conflict-free ds_read_b128
    constexpr int VLD = D + 32;       // V tile row stride: 64 bytes past a multiple of 256 -> conflict-free transpose reads
    static_assert(((D / 8) & 1) == 0, "K row stride");
    static_assert((VLD * 2) % 256 == 64 || (VLD * 2) % 256 == 192, "V row stride");
    constexpr int NLOAD = AT_KT * CH / AT_THREADS;  // chunks per thread, operand and tile
    static_assert(AT_KT * CH % AT_THREADS == 0, "tile chunks per thread");

    extern __shared__ __attribute__((aligned(16))) u16 lds[];
    u16 *const Ks = lds;                                  // [2][AT_KT][KLD]
    u16 *const Vs = lds + 2 * AT_KT * KLD;                // [2][AT_KT][VLD]

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int hh = lane >> 5;                             // lane half
    const int b = blockIdx.z, head = blockIdx.y;
    const long tok = 3L * H * D;                          // elements from one token to the next
    const u16 *const qbase = qkv + (long)b * N * tok + (long)head * D;
    const u16 *const kbase = qbase + (long)H * D;
    const u16 *const vbase = qbase + 2L * H * D;

    // Q^T fragments: lane (query, half) holds Q[query][16 ks + 8 half .. +7] for every k-step
    const int query = blockIdx.x * AT_QB + wave * AT_QW + (lane & 31);
    const int qrow = query < N ? query : N - 1;
    bf16x8 qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8 *>(qbase + (long)qrow * tok + ks * 16 + hh * 8);

    f32x16 o[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[d][i] = 0.0f;
    float m_run = -INFINITY, l_run = 0.0f;

    // tile staging: chunk c of the tile = (key c / CH, 16-byte piece c % CH).  The staged chunks are NAMED registers, not
    // arrays: as `uint4 rk[NLOAD]` they lived in scratch memory (private stack: 32 NLOAD bytes per lane) -- every tile's
    // global loads were waited for on the spot, stored to scratch and read back for the LDS write.
    static_assert(NLOAD >= 1 && NLOAD <= 4, "staging registers");
    uint4 rk0, rk1, rk2, rk3, rv0, rv1, rv2, rv3;
    rk0 = rk1 = rk2 = rk3 = rv0 = rv1 = rv2 = rv3 = make_uint4(0u, 0u, 0u, 0u);
    auto fetch1 = [&](int key0, int i, uint4 &k, uint4 &v) __attribute__((always_inline)) {
        const int c = t + i * AT_THREADS;
        const int key = key0 + c / CH;
        const long off = (long)(key < N ? key : N - 1) * tok + (c % CH) * 8;
        k = *reinterpret_cast<const uint4 *>(kbase + off);
        v = *reinterpret_cast<const uint4 *>(vbase + off);
    };
    auto fetch = [&](int key0) __attribute__((always_inline)) {
        fetch1(key0, 0, rk0, rv0);
        if (NLOAD > 1) fetch1(key0, 1, rk1, rv1);
        if (NLOAD > 2) fetch1(key0, 2, rk2, rv2);
        if (NLOAD > 3) fetch1(key0, 3, rk3, rv3);
    };
    auto stage1 = [&](int buf, int i, const uint4 &k, const uint4 &v) __attribute__((always_inline)) {
        const int c = t + i * AT_THREADS;
        *reinterpret_cast<uint4 *>(&Ks[(buf * AT_KT + c / CH) * KLD + (c % CH) * 8]) = k;
        *reinterpret_cast<uint4 *>(&Vs[(buf * AT_KT + c / CH) * VLD + (c % CH) * 8]) = v;
    };
    auto stage = [&](int buf) __attribute__((always_inline)) {
        stage1(buf, 0, rk0, rv0);
        if (NLOAD > 1) stage1(buf, 1, rk1, rv1);
        if (NLOAD > 2) stage1(buf, 2, rk2, rv2);
        if (NLOAD > 3) stage1(buf, 3, rk3, rv3);
    };
    // transpose-read addressing (see csrc/gemm.hip, gemm_tn): lane 4q+p of a 16-lane group supplies row q,
    // columns 4p..4p+3 of a 4 x 16 block and receives column (lane & 15) of it; groups 0/1 cover columns 0-15 /
    // 16-31 of a 32-wide tile, lane halves the row blocks 4h.. (and 8 rows further for the second four)
    const int grp = lane >> 4, gi = lane & 15;
    const int tr_row = 4 * (grp >> 1) + (gi >> 2);
    const int tr_col = 16 * (grp & 1) + 4 * (gi & 3);
    typedef __attribute__((address_space(3))) s16x4 *lds_ptr;

    const int tiles = (N + AT_KT - 1) / AT_KT;
    fetch(0);
    stage(0);
    __syncthreads();
    for (int j = 0; j < tiles; ++j) {
        const int buf = j & 1;
        if (j + 1 < tiles) fetch((j + 1) * AT_KT);
        const u16 *const Kt = Ks + buf * AT_KT * KLD;
        const u16 *const Vt = Vs + buf * AT_KT * VLD;
        // S^T for both 32-key blocks of the tile at once: two independent accumulator chains keep the matrix pipe issuing
        // (one chain of D/16 dependent MFMAs left it waiting on its own result between instructions), ONE online-softmax
        // step per 64 keys.  (Tried on top and dropped: the second block's products and the first block's V^T P^T placed
        // between the softmax instructions of the other block, one wave overlapping its own matrix and vector work: 366
        // against 364 us at cfg5's shape -- the partner wave of the SIMD already fills those gaps.)
        static_assert(AT_KT == 64, "two 32-key blocks per tile");
        f32x16 s0, s1;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            s0[i] = 0.0f;
            s1[i] = 0.0f;
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 a0 = *reinterpret_cast<const bf16x8 *>(&Kt[(lane & 31) * KLD + ks * 16 + hh * 8]);
            const bf16x8 a1 = *reinterpret_cast<const bf16x8 *>(&Kt[(32 + (lane & 31)) * KLD + ks * 16 + hh * 8]);
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, qf[ks], s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, qf[ks], s1, 0, 0, 0);
        }
        // s0[i] / s1[i] = <k, q> for key = j*64 + {0, 32} + 4*hh + (i&3) + 8*(i>>2), query = this lane's (unscaled)
        if ((j + 1) * AT_KT > N) {   // (workgroup-uniform) the last, ragged tile: keys past the cloud drop out
            const int key_lo = j * AT_KT + 4 * hh;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int key = key_lo + (i & 3) + 8 * (i >> 2);
                s0[i] = key < N ? s0[i] : -INFINITY;
                s1[i] = key + 32 < N ? s1[i] : -INFINITY;
            }
        }
        float mx = fmaxf(s0[0], s1[0]);
#pragma unroll
        for (int i = 1; i < 16; ++i) mx = fmaxf(mx, fmaxf(s0[i], s1[i]));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        // scale > 0: the maximum of the scaled scores is the scaled maximum (rounding is monotonic)
        const float m_new = fmaxf(m_run, mx * scale_log2e);   // finite from the first tile on (key 0 exists)
        const float alpha = m_run == -INFINITY ? 0.0f : __builtin_amdgcn_exp2f(m_run - m_new);
        float psum = 0.0f;
        float p[32];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            // exp2(s * scale - m) with the hardware exponential (arguments <= 0; below -126 it returns 0, as wanted)
            p[i] = __builtin_amdgcn_exp2f(fmaf(s0[i], scale_log2e, -m_new));
            p[16 + i] = __builtin_amdgcn_exp2f(fmaf(s1[i], scale_log2e, -m_new));
            psum += p[i] + p[16 + i];
        }
        psum += __shfl_xor(psum, 32);
        l_run = fmaf(l_run, alpha, psum);
        m_run = m_new;
        // the running maximum settles after the first tiles: rescale only when some query's changed
        if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
            for (int d = 0; d < DT; ++d)
#pragma unroll
                for (int i = 0; i < 16; ++i) o[d][i] *= alpha;
        }
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {   // 16 keys each: blocks 0, 1 of the first 32, 2, 3 of the second
            bf16x8 pf;
#pragma unroll
            for (int e = 0; e < 8; ++e) pf[e] = (short)f2bf(p[blk * 8 + e]);
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                const u16 *pv = &Vt[(blk * 16 + tr_row) * VLD + d * 32 + tr_col];
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)pv);
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(pv + 8 * VLD));
                const bf16x8 a = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, pf, o[d], 0, 0, 0);
            }
        }
        if (j + 1 < tiles) stage(buf ^ 1);
        __syncthreads();
    }
    if (query >= N) return;
    const float inv = 1.0f / l_run;
    u16 *const orow = out + ((long)b * N + query) * ((long)H * D) + (long)head * D;
    // o[d][i] = O[query][32 d + (i&3) + 8 (i>>2) + 4 hh]: four consecutive channels per register quad
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            uint2 w;
            w.x = (uint32_t)f2bf(o[d][4 * g + 0] * inv) | ((uint32_t)f2bf(o[d][4 * g + 1] * inv) << 16);
            w.y = (uint32_t)f2bf(o[d][4 * g + 2] * inv) | ((uint32_t)f2bf(o[d][4 * g + 3] * inv) << 16);
            *reinterpret_cast<uint2 *>(orow + d * 32 + 8 * g + 4 * hh) = w;
        }
}

template <int D>
int launch_attention(const void *qkv, int B, int N, int H, float scale, void *out, hipStream_t st)
{
    constexpr int KLD = D + 8, VLD = D + 32;
    const size_t lds = (size_t)2 * AT_KT * (KLD + VLD) * sizeof(u16);
    static const bool ok = hipFuncSetAttribute((const void *)attention_fwd_bf16_kernel<D>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
    if (!ok) return PCB_ERR_LAUNCH;
    const dim3 grid((unsigned)((N + AT_QB - 1) / AT_QB), (unsigned)H, (unsigned)B);
    hipLaunchKernelGGL(attention_fwd_bf16_kernel<D>, grid, dim3(AT_THREADS), lds, st, (const u16 *)qkv, N, H,
                       scale * 1.4426950408889634f, (u16 *)out);
    return pcb_check_launch();
}

}  // namespace

extern "C" int pcb_attention_fwd_bf16(const void *qkv, int B, int N, int H, int D, float scale, void *out, void *stream)
{
    if (!qkv || !out || B <= 0 || N <= 0 || H <= 0 || !(scale > 0.0f)) return PCB_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    switch (D) {
        case 64: return launch_attention<64>(qkv, B, N, H, scale, out, st);
        case 128: return launch_attention<128>(qkv, B, N, H, scale, out, st);
        case 192: return launch_attention<192>(qkv, B, N, H, scale, out, st);
        case 256: return launch_attention<256>(qkv, B, N, H, scale, out, st);
        default: return PCB_ERR_UNSUPPORTED;
    }
}
