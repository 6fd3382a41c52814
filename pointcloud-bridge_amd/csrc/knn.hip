// kNN graph construction for gfx950 (DGCNN dynamic graph).
//
// Replaces DGCNN.knn, models/DGCNN.py:49-70 of the reference: a [B,N,N] fp32 matrix from a batched
// matmul plus two broadcast adds, then torch.topk over it (2.1 GB per layer at B=8, N=8192).
// Here the matrix never exists.  The inner products run on the fp32-input matrix cores
// (v_mfma_f32_32x32x2_f32): exact fp32, bit for bit the k-ordered fma chain
//     dot = fma(x_{D-1}, y_{D-1}, ... fma(x_1, y_1, x_0*y_0))
// that the CPU oracle and the reference's K=3 sgemm produce, at the fp32 vector peak rate but with
// both operands in registers -- which leaves the VALU free for the selection.
//
//   workgroup = 4 waves = 128 queries of one scene; wave = 32 queries (MFMA column n = lane & 31)
//   candidates are staged 128 (64 for D > 64) at a time through LDS (shared by the 4 waves), 32 per MFMA tile
//   D[m][n]: lane (n, h) receives 16 of the tile's 32 candidates, m = (i&3) + 8(i>>2) + 4h
//   pd(i,j) = (|xi|^2 + (-2*dot)) + |xj|^2                                   (DGCNN.py:63-65)
//   each lane keeps the K best (distance, index) of ITS half of the candidates sorted in registers.
//   Selection is the expensive part on a SIMD machine: a lane rarely has a new top-K member, but a
//   wave would pay the ~100-instruction sorted insertion whenever ANY of its 64 lanes has one.
//   So lanes only APPEND qualifying candidates (distance below their current k-th best) to a small
//   per-lane queue in LDS -- a predicated store -- and the wave drains all queues together when one
//   runs full: the j-th queued entries of all lanes are inserted in the same pass.
//   lanes n and n+32 merge their lists at the end (ties: lower index first)
//   The candidate stages are double-buffered: the next stage's rows are in flight (registers) while
//   the MFMAs of the current one run, one barrier per stage.  |x_j|^2 comes from a small pre-pass
//   (knn_norms_kernel: one lane per point, in the summation order of the reference's torch.sum(x**2))
//   instead of 64 lanes recomputing it per stage.
#include "pcb_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kThreads = 256;

__device__ __forceinline__ int wave_max_i(int v)
{
    v = max(v, dpp_i<PCB_ROW_ROR(8)>(v));
    v = max(v, dpp_i<PCB_ROW_ROR(4)>(v));
    v = max(v, dpp_i<PCB_ROW_ROR(2)>(v));
    v = max(v, dpp_i<PCB_ROW_ROR(1)>(v));
    return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

// insertion of (d, cand) into a list sorted ascending by (d, then insertion order)
template <int K>
__device__ __forceinline__ void insert_sorted(float (&bd)[K], int (&bi)[K], float d, int cand)
{
    bool hi = true;  // d < bd[s] for the slot being written
#pragma unroll
    for (int s = K - 1; s > 0; --s) {
        const bool lo = d < bd[s - 1];
        bd[s] = lo ? bd[s - 1] : (hi ? d : bd[s]);
        bi[s] = lo ? bi[s - 1] : (hi ? cand : bi[s]);
        hi = lo;
    }
    bd[0] = hi ? d : bd[0];
    bi[0] = hi ? cand : bi[0];
}

// same, ordered by (d, index): used when two lanes' lists are merged
template <int K>
__device__ __forceinline__ void insert_sorted_by_index(float (&bd)[K], int (&bi)[K], float d, int cand)
{
    bool hi = d < bd[K - 1] || (d == bd[K - 1] && cand < bi[K - 1]);
    if (!hi) return;
#pragma unroll
    for (int s = K - 1; s > 0; --s) {
        const bool lo = d < bd[s - 1] || (d == bd[s - 1] && cand < bi[s - 1]);
        bd[s] = lo ? bd[s - 1] : (hi ? d : bd[s]);
        bi[s] = lo ? bi[s - 1] : (hi ? cand : bi[s]);
        hi = lo;
    }
    bd[0] = hi ? d : bd[0];
    bi[0] = hi ? cand : bi[0];
}

// |x|^2 of every point in the order of the reference's torch.sum(x**2, dim=2) (models/DGCNN.py:64), multiply and
// add rounded separately.  ATen's vectorised reduction over a contiguous axis (tools/sgemm_order.py, bit-identical
// for D = 32..256): D % 32 == 0 -> four 8-lane accumulators take the 8-channel chunks in turn, accumulators added
// left to right, then the lanes left to right; other D (D = 3: the coordinates) -> one left-to-right chain.
__global__ __launch_bounds__(kThreads) void knn_norms_kernel(const float *__restrict__ x, long rows, int D,
                                                              float *__restrict__ norms)
{
    const long r = (long)blockIdx.x * kThreads + threadIdx.x;
    if (r >= rows) return;
    const float *__restrict__ p = x + r * D;
    float s = 0.0f;
    if (D % 32 == 0) {
        for (int l = 0; l < 8; ++l) {
            float v = 0.0f;
            for (int u = 0; u < 4; ++u) {
                float a = __fmul_rn(p[u * 8 + l], p[u * 8 + l]);
                for (int c0 = u * 8 + 32; c0 < D; c0 += 32) a = __fadd_rn(a, __fmul_rn(p[c0 + l], p[c0 + l]));
                v = u ? __fadd_rn(v, a) : a;
            }
            s = l ? __fadd_rn(s, v) : v;
        }
    } else {
        for (int c = 0; c < D; ++c) {
            const float v = p[c];
            s = c ? __fadd_rn(s, __fmul_rn(v, v)) : __fmul_rn(v, v);
        }
    }
    norms[r] = s;
}

// DP = D rounded up to a power of two >= 4; channels D..DP-1 are zero padding, which changes no
// rounding step of the chain (fma(0, 0, acc) == acc) nor of the norms (s + 0*0 == s).
template <int DP, int K>
__global__ __launch_bounds__(kThreads, 2) void knn_mfma_kernel(const float *__restrict__ x,
                                                             const float *__restrict__ norms, int N, int D,
                                                             int k, int64_t *__restrict__ out,
                                                             const int *__restrict__ only_if, int only_if_stride,
                                                             const int *__restrict__ qlist = nullptr,
                                                             const int *__restrict__ qcount = nullptr, int qlist_min = 0,
                                                             const int *__restrict__ redo = nullptr)
{
    // optional per-scene switch (csrc/knngrid.hip serves the other scenes): uniform per workgroup
    if (only_if && only_if[(size_t)blockIdx.y * only_if_stride] == 0) return;
    // optional query list (the screened kernel's uncertified queries): qlist[b*N + slot], qcount[b] of them
    // (up to kPerQueryMax of them are served by knn_query_kernel instead, when qlist_min > 0)
    // -- unless that kernel flagged the scene (redo: a candidate list of its overflowed)
    if (qlist && ((int)blockIdx.x * 128 >= qcount[blockIdx.y] ||
                  (qcount[blockIdx.y] <= qlist_min && !(redo && redo[blockIdx.y]))))
        return;
    constexpr int S = DP / 2;         // MFMA steps per tile (two channels each)
    constexpr int LD = DP + 4;        // LDS row stride in floats: rows 16 B apart in bank space
    constexpr int kTC = DP <= 32 ? 128 : (DP <= 64 ? 64 : 32);  // candidates per LDS stage (<= 17.4 KB)
    constexpr int QCAP = 20;          // queue slots per lane; drained when a lane has > QCAP - 16
    constexpr int NPT = kTC * DP / kThreads;  // staged elements per thread and stage
    __shared__ __attribute__((aligned(16))) float tiles[2][kTC * LD];  // row m: [h][s] = x[m][2s+h]
    __shared__ __attribute__((aligned(16))) float cnrms[2][kTC];
    __shared__ float qd[QCAP][kThreads];  // [slot][thread]: a lane's slots are a bank-conflict-free column
    __shared__ int qi[QCAP][kThreads];

    const int b = blockIdx.y;
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int h = lane >> 5;
    const float *__restrict__ xb = x + (size_t)b * N * D;

    // this lane's query (two lanes per query: h = 0 holds the even channels, h = 1 the odd ones)
    const int qslot = blockIdx.x * 128 + wave * 32 + (lane & 31);
    const int qn_idx = qlist ? (qslot < qcount[blockIdx.y] ? qlist[(size_t)blockIdx.y * N + qslot] : N) : qslot;
    const bool qvalid = qn_idx < N;
    const float *__restrict__ xq = xb + (size_t)(qvalid ? qn_idx : N - 1) * D;
    float q[S];
#pragma unroll
    for (int s = 0; s < S; ++s) q[s] = (2 * s + h) < D ? xq[2 * s + h] : 0.0f;
    const float qnorm = norms[(size_t)b * N + (qvalid ? qn_idx : N - 1)];

    float bd[K];
    int bi[K];
#pragma unroll
    for (int s = 0; s < K; ++s) {
        bd[s] = INFINITY;
        bi[s] = 0x7fffffff;
    }
    int qcnt = 0;
    // drain: every lane inserts its j-th queued candidate in the same pass (queue order = index order)
    auto drain = [&]() {
        const int mx = __builtin_amdgcn_readfirstlane(wave_max_i(qcnt));
        for (int j = 0; j < mx; ++j) {
            if (j < qcnt) {
                const float dv = qd[j][t];
                const int ci = qi[j][t];
                if (dv < bd[K - 1]) insert_sorted<K>(bd, bi, dv, ci);
            }
        }
        qcnt = 0;
    };

    // stage = kTC candidate rows, channels de-interleaved into [evens | odds]; a thread's NPT
    // elements are e = t + i*256 (consecutive lanes -> consecutive channels of a row: coalesced)
    float pre[NPT];
    float pre_n = 0.0f;
    auto fetch = [&](int base) {
        if (D == DP && base + kTC <= N) {
            // whole stage inside the cloud, rows DP wide: the stage is one contiguous block and a
            // thread's elements sit at fixed strides from one (wave-uniform + lane) address
            const float *__restrict__ sp = xb + (size_t)base * DP + t;
#pragma unroll
            for (int i = 0; i < NPT; ++i) pre[i] = sp[i * kThreads];
        } else {
#pragma unroll
            for (int i = 0; i < NPT; ++i) {
                const int e = t + i * kThreads;
                const int m = e / DP, c = e % DP;
                const int row = base + m;
                pre[i] = (row < N && c < D) ? xb[(size_t)row * D + c] : 0.0f;
            }
        }
        if (t < kTC) pre_n = base + t < N ? norms[(size_t)b * N + base + t] : 0.0f;
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const int e = t + i * kThreads;
            const int m = e / DP, c = e % DP;
            tiles[buf][m * LD + (c & 1) * S + (c >> 1)] = pre[i];
        }
        if (t < kTC) cnrms[buf][t] = pre_n;
    };
    fetch(0);
    stash(0);
    __syncthreads();
    int buf = 0;
    for (int base = 0; base < N; base += kTC, buf ^= 1) {
        const bool more = base + kTC < N;
        if (more) fetch(base + kTC);  // in flight under this stage's MFMAs
        const float *const tile = tiles[buf];
        const float *const cnrm = cnrms[buf];

#pragma unroll 1
        for (int tt = 0; tt < kTC / 32; ++tt) {
            if (base + tt * 32 >= N) break;  // wave-uniform
            // A operand: lane (m, h) holds x_cand[m][2s + h], s = 0..S-1
            float a[S];
            const float *row = &tile[(tt * 32 + (lane & 31)) * LD + h * S];
#pragma unroll
            for (int s4 = 0; s4 < S / 4; ++s4) {
                const float4 v = *reinterpret_cast<const float4 *>(row + s4 * 4);
                a[s4 * 4 + 0] = v.x;
                a[s4 * 4 + 1] = v.y;
                a[s4 * 4 + 2] = v.z;
                a[s4 * 4 + 3] = v.w;
            }
            if (S < 4) {
#pragma unroll
                for (int s = 0; s < S; ++s) a[s] = row[s];
            }
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
#pragma unroll
            for (int s = 0; s < S; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], q[s], acc, 0, 0, 0);

            // distances of this lane's 16 candidates; those below the lane's current k-th best are
            // appended to its queue (the threshold is stale until the next drain: a few extra
            // entries, never a missed one)
            // (the queues are drained when one of them could overflow in the next group of four: the wave pays a drain
            // pass for its FULLEST queue, so late drains keep more lanes busy per pass)
            float thr = bd[K - 1];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 cn = *reinterpret_cast<const float4 *>(&cnrm[tt * 32 + 8 * g + 4 * h]);
                const float cnv[4] = {cn.x, cn.y, cn.z, cn.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int cand = base + tt * 32 + 8 * g + 4 * h + j;
                    // (xx + inner) + xx^T with inner = -2*dot (exact), DGCNN.py:63-65
                    const float v = __fadd_rn(__fmaf_rn(-2.0f, acc[g * 4 + j], qnorm), cnv[j]);
                    if (cand < N && v < thr) {
                        qd[qcnt][t] = v;
                        qi[qcnt][t] = cand;
                        ++qcnt;
                    }
                }
                if (__any(qcnt > QCAP - 4)) {
                    drain();
                    thr = bd[K - 1];
                }
            }
        }
        if (more) stash(buf ^ 1);  // the other buffer was last read one barrier ago
        __syncthreads();
    }
    drain();

    // lanes n and n+32 saw disjoint halves of the candidates: merge (ties: lower index first)
#pragma unroll
    for (int s = 0; s < K; ++s) {
        const float od = __shfl(bd[s], (lane & 31) + 32);
        const int oi = __shfl(bi[s], (lane & 31) + 32);
        if (h == 0) insert_sorted_by_index<K>(bd, bi, od, oi);
    }
    if (h == 0 && qvalid) {
        int64_t *__restrict__ o = out + ((size_t)b * N + qn_idx) * k;
#pragma unroll
        for (int s = 0; s < K; ++s)
            if (s < k) o[s] = (int64_t)bi[s];
    }
}

// ---- feature-space kNN with a split-bf16 screening pass (D >= 32: DGCNN's second to fourth graphs) ----------------------
// The exact kernel above is bound by the fp32 matrix core (v_mfma_f32_32x32x2_f32: 1/16 of the bf16 rate) IN SERIES with
// the selection's vector instructions (the fp32 MFMA does not co-execute with VALU work, and its LDS footprint leaves one
// wave per SIMD).  Here the N x N products run on v_mfma_f32_32x32x16_bf16 over a two-term bf16 split of the coordinates,
//     x = xh + xl + r,  xh = bf16(x), xl = bf16(x - xh), |r| <= 2^-16 |x|,      dot' = xh.yh + xh.yl + xl.yh
// (three MFMAs per 16 channels: 5.3x fewer matrix cycles than fp32, and they overlap the partner wave's selection).
// dot' differs from the real inner product by xl.yl + r.y + x.r': at most 3 * 2^-16 sum|x_c y_c|; the fp32 accumulation
// of the 3 D products (any order, any rounding mode) adds at most 3 D 2^-23 sum|x_c y_c|, the reference's own fma chain
// D 2^-24 sum|x_c y_c|; with sum|x_c y_c| <= (|x|^2 + |y|^2)/2 and v = (qn - 2 dot) + cn the approximate distance v'
// (same norms, same expression) satisfies
//     |v' - v| <= c (|x_i|^2 + |x_j|^2),   c = 3*2^-16 + 3 D 2^-23 + D 2^-24   (7.3e-5 for D = 64, 9.9e-5 for D = 128)
// so key_j = v'_j - c |x_j|^2 (computed as (qn - 2 dot') + |x_j|^2 (1 - c): no extra instruction) is a lower bound of
// v_j + c qn.  Per query the KS smallest keys are kept (the same queue / sorted-list machinery as above); then
//     * the distances of the kept candidates are recomputed EXACTLY -- the k-ordered fp32 fma chain of the reference's
//       sgemm, the expression (qn + (-2 dot)) + cn of models/DGCNN.py:63-65 -- and sorted by (distance, index);
//     * the result is CERTIFIED: every candidate outside the kept set has a key >= the largest kept key, hence a true
//       distance >= that - c qn; if this exceeds the k-th exact distance of the kept set, nobody outside can enter the
//       top k (not even through a tie) and the k best of the set ARE the reference's list, in the reference's order;
//     * a query that cannot be certified (ties at the boundary, duplicated points, neighbours packed closer than the
//       bound) goes on a list with T = the k-th exact distance of its kept set; knn_query_kernel recomputes it exactly
//       among the candidates with a distance <= T (one workgroup per query), and scenes with more than kPerQueryMax such
//       queries go through the exact kernel above (its list mode).
// Output bit-identical to pcb_knn for every input; how much faster depends on how many queries certify.
// Measured (B = 8, N = 8192, k = 20, tools/knn_screen_bench.py): D = 64 1.05 ms -> 0.60 ms per call (screen kernel
// 0.53 ms, 7 of 65536 queries recomputed on Gaussian features, 0-8 on DGCNN's own), D = 128 1.58 -> 0.96 ms; a cloud
// whose norms dwarf its neighbour distances (x + 20) certifies nothing and costs 1.5x the exact kernel.  The screen
// kernel is bound by vector-instruction issue: 191 VALU + 105 scalar instructions per 32 x 32 tile and wave against
// 12 MFMA (s_memtime: 2560 cycles per tile with two waves per SIMD), i.e. by the selection, not by the products.
typedef short bf16x8s __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned short f2bf_rn(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
__device__ __forceinline__ float bf2f(unsigned short u) { return __uint_as_float((uint32_t)u << 16); }

// xh / xl [rows][DP] (zero beyond D)
__global__ __launch_bounds__(kThreads) void knn_split_kernel(const float *__restrict__ x, long rows, int D, int DP,
                                                              unsigned short *__restrict__ xh,
                                                              unsigned short *__restrict__ xl)
{
    const long e = (long)blockIdx.x * kThreads + threadIdx.x;
    if (e >= rows * DP) return;
    const long r = e / DP;
    const int c = (int)(e % DP);
    const float v = c < D ? x[r * D + c] : 0.0f;
    const unsigned short hi = f2bf_rn(v);
    xh[e] = hi;
    xl[e] = f2bf_rn(v - bf2f(hi));   // v - hi is exact in fp32
}

// The kept set lives in registers as an ascending list of PACKED keys: the order-preserving integer image of the key with
// its low `idx_bits` bits replaced by the candidate index (the truncation only lowers a key: the bound stays a bound, at
// 2^-(23 - idx_bits) of the distance).  One 32-bit word per entry makes the insertion one v_med3_u32 per slot
// (b[s-1] <= b[s], so the new b[s] is the median of b[s-1], b[s] and x; top slot first, every slot still sees its old
// lower neighbour), a fifth of the compare-and-select chain of the exact kernel's (distance, index) pairs, and halves
// the LDS queues.
__device__ __forceinline__ uint32_t med3u(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
template <int KS>
__device__ __forceinline__ void insert_packed(uint32_t (&b)[KS], uint32_t x)
{
#pragma unroll
    for (int s = KS - 1; s > 0; --s) b[s] = med3u(b[s - 1], b[s], x);
    b[0] = min(b[0], x);
}
__device__ __forceinline__ uint32_t sortable(float v)
{
    const uint32_t u = __float_as_uint(v);
    return u ^ ((uint32_t)((int)u >> 31) | 0x80000000u);
}
__device__ __forceinline__ float unsortable(uint32_t s) { return __uint_as_float(s ^ ((s >> 31) ? 0x80000000u : 0xffffffffu)); }

template <int DP, int K /* k rounded up: 8, 20 */, int KS /* candidates kept by the screening pass */>
__global__ __launch_bounds__(kThreads, 2) void knn_screen_kernel(const float *__restrict__ x,
                                                               const unsigned short *__restrict__ xh,
                                                               const unsigned short *__restrict__ xl,
                                                               const float *__restrict__ norms, int nscenes, int N, int D,
                                                               int k, int idx_bits, int64_t *__restrict__ out,
                                                               int *__restrict__ qlist, float *__restrict__ qtk,
                                                               int *__restrict__ qcount)
{
    constexpr int S = DP / 16;                   // MFMA steps per tile (16 channels each), three MFMAs per step
    constexpr int M = 3 * S;
    constexpr int LD = DP + 8;                   // LDS row stride in bf16: successive rows 4 banks apart (b128 reads)
    constexpr int kTC = DP <= 64 ? 64 : 32;      // candidates per LDS stage: two workgroups per CU (80 KB each)
    constexpr int T = kTC / 32;                  // tiles per stage
    constexpr int QCAP = 32;
    constexpr int kMergeTiles = 32;  // tiles between two hand-overs of the half-lane lists (8 / 16 / 32 / 64: 603 / 583 / 582 / 580 us)
    constexpr int NV = kTC * DP / 8 / kThreads;  // 16-byte loads per thread, stage and half (hi / lo)
    static_assert(NV >= 1, "stage smaller than the workgroup");
    __shared__ __attribute__((aligned(16))) unsigned short tiles[2][2][kTC * LD];   // [buffer][hi | lo]
    __shared__ __attribute__((aligned(16))) float cnrms[3][kTC];   // a ring of three: read one barrier longer than the tiles
    __shared__ uint32_t qk[QCAP][kThreads];

    // workgroups go to the 8 XCDs round-robin by their linear index: give every XCD whole scenes (consecutive query
    // blocks of one scene), so that the scene's split coordinates -- read by all of its query blocks -- stay in ONE 4 MB L2
    // instead of passing through all eight (grid = 8 * ceil(blocks / 8), surplus workgroups leave)
    const int qblocks = (N + 127) / 128;
    const int per_xcd = gridDim.x / 8;
    const int lin = ((int)blockIdx.x % 8) * per_xcd + (int)blockIdx.x / 8;
    if (lin >= qblocks * nscenes) return;
    const int b = lin / qblocks, qblock = lin % qblocks;
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int h = lane >> 5;
    const float *__restrict__ xb = x + (size_t)b * N * D;
    const unsigned short *__restrict__ xhb = xh + (size_t)b * N * DP;
    const unsigned short *__restrict__ xlb = xl + (size_t)b * N * DP;
    const float *__restrict__ nb = norms + (size_t)b * N;
    const uint32_t imask = (1u << idx_bits) - 1u;

    const int qn_idx = qblock * 128 + wave * 32 + (lane & 31);
    const bool qvalid = qn_idx < N;
    const int qrow = qvalid ? qn_idx : N - 1;
    // B operand: lane (n, h) holds channels 16 s + 8 h .. + 8 of query n
    bf16x8s qh[S], ql[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        qh[s] = *reinterpret_cast<const bf16x8s *>(xhb + (size_t)qrow * DP + 16 * s + 8 * h);
        ql[s] = *reinterpret_cast<const bf16x8s *>(xlb + (size_t)qrow * DP + 16 * s + 8 * h);
    }
    const float qnorm = nb[qrow];
    // c of the header comment for this D, with 5 % to spare for the roundings of the key itself
    constexpr float kC = 1.05f * (3.0f / 65536.0f + (3.0f * DP + 8.0f) / 8388608.0f + DP / 16777216.0f);

    uint32_t bd[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) bd[s] = 0xffffffffu;
    int qcnt = 0;
    // every lane inserts its queue, slot by slot; lanes past their count insert the neutral word
    auto drain = [&]() {
        const int mx = __builtin_amdgcn_readfirstlane(wave_max_i(qcnt));
        uint32_t w = qk[0][t];
        for (int j = 0; j < mx; ++j) {
            const uint32_t wn = qk[j + 1 < QCAP ? j + 1 : QCAP - 1][t];
            insert_packed<KS>(bd, j < qcnt ? w : 0xffffffffu);
            w = wn;
        }
        qcnt = 0;
    };
    // Lanes n and n+32 serve the same query on disjoint halves of the candidates.  Every kMergeTiles tiles the upper
    // lane hands its list to the lower one (which then holds the best KS of everything seen so far), empties its own and
    // keeps the merged list's last entry as a CAP: a key above it cannot be among the best KS of the union, whichever
    // half it comes from.  The filter threshold of both lanes is then the union's KS-th key instead of a half's -- about
    // half as many candidates pass, half as many queue entries to insert -- and the two lists never share an entry.
    uint32_t cap = 0xffffffffu;
    // keys up to this value may still enter the list (finite: a candidate beyond the cloud has an infinite key)
    auto threshold = [&]() { return unsortable(min(min(bd[KS - 1], cap) | imask, 0xff7fffffu)); };
    auto merge_halves = [&]() {
        uint32_t od[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) od[s] = (uint32_t)__shfl((int)bd[s], lane ^ 32);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            od[s] = h ? 0xffffffffu : od[s];   // the upper lane inserts neutral words into its emptied list
            bd[s] = h ? 0xffffffffu : bd[s];
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) insert_packed<KS>(bd, od[s]);
        const uint32_t merged_worst = (uint32_t)__shfl((int)bd[KS - 1], lane & 31);
        cap = h ? merged_worst : cap;
    };

    // staged chunks in NAMED registers (as `uint4 pre[2][NV]` behind conditional loads they lived in scratch memory:
    // every stage's global loads were waited for on the spot, stored to the private stack and read back for the LDS write)
    static_assert(NV >= 1 && NV <= 2, "staging registers");
    uint4 ph0, ph1, pl0, pl1;
    ph0 = ph1 = pl0 = pl1 = make_uint4(0u, 0u, 0u, 0u);
    float pre_n = 0.0f;
    auto fetch1 = [&](int base, int i, uint4 &vh, uint4 &vl) __attribute__((always_inline)) {
        const int e8 = t + i * kThreads;
        const int m = e8 / (DP / 8), c = (e8 % (DP / 8)) * 8;
        const int row = base + m < N ? base + m : N - 1;   // (rows past the cloud: any finite data -- their key term is infinite)
        vh = *reinterpret_cast<const uint4 *>(xhb + (size_t)row * DP + c);
        vl = *reinterpret_cast<const uint4 *>(xlb + (size_t)row * DP + c);
    };
    auto fetch = [&](int base) __attribute__((always_inline)) {
        fetch1(base, 0, ph0, pl0);
        if (NV > 1) fetch1(base, 1, ph1, pl1);
        // (no arithmetic on the loaded value here: it would wait for the loads of the whole stage)
        if (t < kTC) pre_n = base + t < N ? nb[base + t] : INFINITY;
    };
    auto stash1 = [&](int buf, int i, const uint4 &vh, const uint4 &vl) __attribute__((always_inline)) {
        const int e8 = t + i * kThreads;
        const int m = e8 / (DP / 8), c = (e8 % (DP / 8)) * 8;
        *reinterpret_cast<uint4 *>(&tiles[buf][0][m * LD + c]) = vh;
        *reinterpret_cast<uint4 *>(&tiles[buf][1][m * LD + c]) = vl;
    };
    auto stash = [&](int buf, int ring) __attribute__((always_inline)) {
        stash1(buf, 0, ph0, pl0);
        if (NV > 1) stash1(buf, 1, ph1, pl1);
        // |x_j|^2 (1 - c): the key's candidate term; a candidate beyond the cloud gets an infinite one and never passes
        if (t < kTC) cnrms[ring][t] = pre_n * (1.0f - kC);
    };
    const int arow = (lane & 31) * LD + 8 * h;   // this lane's A-operand row inside a tile
    // selection over the finished tile `acc` (candidates cbase ..+32, their key terms at cn) with the matrix work of the
    // NEXT tile (nth / ntl, into nacc) issued between the candidates: one wave keeps both pipes busy
    auto step = [&](const f32x16 &acc, const float *cn, int cbase, bool has_next, const unsigned short *nth,
                    const unsigned short *ntl, f32x16 &nacc) {
        bf16x8s ah[S], al[S];
        if (has_next) {
#pragma unroll
            for (int s = 0; s < S; ++s) {
                ah[s] = *reinterpret_cast<const bf16x8s *>(&nth[arow + 16 * s]);
                al[s] = *reinterpret_cast<const bf16x8s *>(&ntl[arow + 16 * s]);
            }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) nacc[i] = 0.0f;
        float4 cn4[4];   // all four reads up front: their latency hides behind the first products
#pragma unroll
        for (int g = 0; g < 4; ++g) cn4[g] = *reinterpret_cast<const float4 *>(&cn[8 * g + 4 * h]);
        float thr = threshold();
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float cnv[4] = {cn4[g].x, cn4[g].y, cn4[g].z, cn4[g].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (has_next) {
#pragma unroll
                    for (int m = (g * 4 + j) * M / 16; m < (g * 4 + j + 1) * M / 16; ++m) {
                        const int s = m / 3;
                        nacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(m % 3 == 0 ? al[s] : ah[s], m % 3 == 1 ? ql[s] : qh[s],
                                                                       nacc, 0, 0, 0);
                    }
                }
                const float v = __fadd_rn(__fmaf_rn(-2.0f, acc[g * 4 + j], qnorm), cnv[j]);   // the key
                if (v <= thr) {
                    qk[qcnt][t] = (sortable(v) & ~imask) | (uint32_t)(cbase + 8 * g + 4 * h + j);
                    ++qcnt;
                }
            }
            if (__any(qcnt > QCAP - 4)) {
                drain();
                thr = threshold();
            }
        }
    };

    f32x16 acc, nacc;
    fetch(0);
    stash(0, 0);
    __syncthreads();
    if (kTC < N) fetch(kTC);
    {   // the first tile has nobody to hide behind
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const bf16x8s a_h = *reinterpret_cast<const bf16x8s *>(&tiles[0][0][arow + 16 * s]);
            const bf16x8s a_l = *reinterpret_cast<const bf16x8s *>(&tiles[0][1][arow + 16 * s]);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_l, qh[s], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_h, ql[s], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_h, qh[s], acc, 0, 0, 0);
        }
    }
    // stage i: tiles in buffer i % 2, key terms in ring slot i % 3.  Per stage ONE barrier, placed before the selection of
    // the stage's last tile: everything read from tiles[i % 2] has been issued by then (the last tile's products ran during
    // the selection of the tile before), so the next stage may be stashed into the other buffer and its first tile's
    // products run behind the last selection; the key terms of the last tile are read after the barrier, hence the ring.
    int buf = 0, ring = 0;
    for (int base = 0; base < N; base += kTC, buf ^= 1, ring = ring == 2 ? 0 : ring + 1) {
        const bool more = base + kTC < N;
#pragma unroll
        for (int tt = 0; tt + 1 < T; ++tt) {
            step(acc, &cnrms[ring][tt * 32], base + tt * 32, true, &tiles[buf][0][(tt + 1) * 32 * LD],
                 &tiles[buf][1][(tt + 1) * 32 * LD], nacc);
            acc = nacc;
        }
        if (more) stash(buf ^ 1, ring == 2 ? 0 : ring + 1);
        __syncthreads();
        if (base + 2 * kTC < N) fetch(base + 2 * kTC);
        step(acc, &cnrms[ring][(T - 1) * 32], base + (T - 1) * 32, more, &tiles[buf ^ 1][0][0], &tiles[buf ^ 1][1][0], nacc);
        acc = nacc;
        if (((base / kTC + 1) % (kMergeTiles / T)) == 0 && more) {   // wave-uniform; queues empty before the hand-over
            drain();
            merge_halves();
        }
    }
    drain();

    // lanes n and n+32 saw disjoint halves of the candidates: both end up with the KS smallest of the union
    {
        uint32_t od[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) od[s] = (uint32_t)__shfl((int)bd[s], lane ^ 32);
#pragma unroll
        for (int s = 0; s < KS; ++s) insert_packed<KS>(bd, od[s]);
    }
    // every candidate outside the kept set has a packed key >= this one (all ones: the set holds the whole cloud)
    const uint32_t worst = bd[KS - 1];

    // exact distances of the kept candidates, lane h those of the slots s = h (mod 2): the k-ordered fma chain of
    // models/DGCNN.py:60-65
    const float *__restrict__ xq = xb + (size_t)qrow * D;
    float ed[KS / 2];
#pragma unroll
    for (int u = 0; u < KS / 2; ++u) {
        const uint32_t pk = h ? bd[2 * u + 1] : bd[2 * u];
        const int cand = (int)(pk & imask);
        float v = INFINITY;
        if (pk != 0xffffffffu) {
            const float *__restrict__ xc = xb + (size_t)cand * D;
            float dot;
            if ((D & 3) == 0) {   // 16-byte loads: a lane's row is contiguous, the chain order is unchanged
                const float4 a0 = *reinterpret_cast<const float4 *>(xq), c0 = *reinterpret_cast<const float4 *>(xc);
                dot = __fmul_rn(a0.x, c0.x);
                dot = __fmaf_rn(a0.y, c0.y, dot);
                dot = __fmaf_rn(a0.z, c0.z, dot);
                dot = __fmaf_rn(a0.w, c0.w, dot);
                for (int c = 4; c < D; c += 4) {
                    const float4 a4 = *reinterpret_cast<const float4 *>(xq + c);
                    const float4 c4 = *reinterpret_cast<const float4 *>(xc + c);
                    dot = __fmaf_rn(a4.x, c4.x, dot);
                    dot = __fmaf_rn(a4.y, c4.y, dot);
                    dot = __fmaf_rn(a4.z, c4.z, dot);
                    dot = __fmaf_rn(a4.w, c4.w, dot);
                }
            } else {
                dot = __fmul_rn(xq[0], xc[0]);
                for (int c = 1; c < D; ++c) dot = __fmaf_rn(xq[c], xc[c], dot);
            }
            v = __fadd_rn(__fmaf_rn(-2.0f, dot, qnorm), nb[cand]);
        }
        ed[u] = v;
    }
    // every lane sorts all KS (exact distance, index) pairs into its K best
    float kd[K];
    int ki[K];
#pragma unroll
    for (int s = 0; s < K; ++s) {
        kd[s] = INFINITY;
        ki[s] = 0x7fffffff;
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const float other = __shfl(ed[s / 2], lane ^ 32);
        const float v = (s & 1) == h ? ed[s / 2] : other;
        insert_sorted_by_index<K>(kd, ki, v, bd[s] == 0xffffffffu ? 0x7fffffff : (int)(bd[s] & imask));
    }
    if (h == 0 && qvalid) {
        float tk = kd[0];   // k <= K
#pragma unroll
        for (int s = 1; s < K; ++s)
            if (s < k) tk = kd[s];
        // the kept keys' floor is a lower bound of every outside key; minus c qn of every outside distance
        const float outside = unsortable(worst & ~imask) - kC * qnorm;
        const bool certified = worst == 0xffffffffu || outside > tk;
        int64_t *__restrict__ o = out + ((size_t)b * N + qn_idx) * k;
#pragma unroll
        for (int s = 0; s < K; ++s)
            if (s < k) o[s] = (int64_t)ki[s];
        if (!certified) {
            const int slot = atomicAdd(&qcount[b], 1);
            qlist[(size_t)b * N + slot] = qn_idx;
            qtk[(size_t)b * N + slot] = tk;   // the true k-th distance is at most this one
        }
    }
}

template <int DP>
int launch_knn(const float *x, float *norms, int B, int N, int D, int k, int64_t *out, hipStream_t st,
               const int *only_if = nullptr, int stride = 0)
{
    const long rows = (long)B * N;
    hipLaunchKernelGGL(knn_norms_kernel, dim3((unsigned)((rows + kThreads - 1) / kThreads)), dim3(kThreads), 0, st, x,
                       rows, D, norms);
    const dim3 grid((N + 127) / 128, B);
    if (k <= 8)
        hipLaunchKernelGGL((knn_mfma_kernel<DP, 8>), grid, dim3(kThreads), 0, st, x, norms, N, D, k, out, only_if, stride);
    else if (k <= 20)
        hipLaunchKernelGGL((knn_mfma_kernel<DP, 20>), grid, dim3(kThreads), 0, st, x, norms, N, D, k, out, only_if, stride);
    else
        hipLaunchKernelGGL((knn_mfma_kernel<DP, 32>), grid, dim3(kThreads), 0, st, x, norms, N, D, k, out, only_if, stride);
    if (!only_if) pcb_account(4.0 * (double)D * N * B + 8.0 * (double)N * k * B);  // (the flagged form rides on pcb_knn_xyz's count)
    return pcb_check_launch();
}


// A FEW uncertified queries: one workgroup per query instead of the list mode of knn_mfma_kernel (whose workgroup scans
// the whole cloud in sequence: 0.8 ms of latency at N = 8192 however few queries it carries).  The screening pass hands
// over T, the k-th exact distance of its kept set: an upper bound of the true k-th distance, so the answer lies among the
// candidates with an exact distance <= T -- usually k of them or a few more.  Every lane computes the exact distances of
// its candidates (same chain, same expression; rows reach the lanes through a wave-private LDS tile, 64 rows x 32
// channels at a time, so that the global loads stay coalesced), candidates <= T go on a short list, and a rank sort by
// (distance, index) writes the first k.  A list that overflows (hundreds of exact ties) flags its scene for the list
// mode.  Serves up to kPerQueryMax queries per scene; beyond that the list mode runs.
constexpr int kPerQueryMax = 1024;
constexpr int kQueryThreads = 512;
constexpr int kQueryList = 512;

__global__ __launch_bounds__(kQueryThreads) void knn_query_kernel(const float *__restrict__ x,
                                                                   const float *__restrict__ norms, int N, int D, int k,
                                                                   int64_t *__restrict__ out, const int *__restrict__ qlist,
                                                                   const float *__restrict__ qtk,
                                                                   const int *__restrict__ qcount, int *__restrict__ redo)
{
    constexpr int W = kQueryThreads / 64, CH = 32, TS = CH + 4;   // row stride 36 floats: 16-byte rows, b128 reads conflict-free
    __shared__ __attribute__((aligned(16))) float tile[W][64][TS];
    __shared__ __attribute__((aligned(16))) float xq[128];
    __shared__ float list_v[kQueryList];
    __shared__ int list_j[kQueryList];
    __shared__ int list_n;
    const int b = blockIdx.y, t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int count = qcount[b];
    if ((int)blockIdx.x >= count || count > kPerQueryMax) return;
    const int q = qlist[(size_t)b * N + blockIdx.x];
    const float tk = qtk[(size_t)b * N + blockIdx.x];
    const float *__restrict__ xb = x + (size_t)b * N * D;
    const float *__restrict__ nb = norms + (size_t)b * N;
    if (t < D) xq[t] = xb[(size_t)q * D + t];
    if (t == 0) list_n = 0;
    __syncthreads();
    const float qnorm = nb[q];
    const bool vec = (D & 3) == 0;
    auto push = [&](int row, float dot) {
        const float v = __fadd_rn(__fmaf_rn(-2.0f, dot, qnorm), nb[row]);
        if (v <= tk) {
            const int slot = atomicAdd(&list_n, 1);
            if (slot < kQueryList) {
                list_v[slot] = v;
                list_j[slot] = row;
            }
        }
    };
    if (vec) {
        // (group of 512 rows, chunk of 32 channels) in sequence, the next one's rows in flight in registers
        const int nchunk = (D + CH - 1) / CH;
        const int total = (N + kQueryThreads - 1) / kQueryThreads * nchunk;   // workgroup-uniform: the barriers are safe
        float4 cur[8], nxt[8];
        auto gload = [&](int it, float4(&r)[8]) {
            const int rb = (it / nchunk) * kQueryThreads + w * 64, c0 = (it % nchunk) * CH;
            const int c = c0 + (lane & 7) * 4;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = rb + i * 8 + (lane >> 3);
                r[i] = (c < D && row < N) ? *reinterpret_cast<const float4 *>(xb + (size_t)row * D + c)
                                          : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
        };
        gload(0, cur);
        float dot = 0.0f;
        for (int it = 0; it < total; ++it) {
            const int c0 = (it % nchunk) * CH;
            const int cw = D - c0 < CH ? D - c0 : CH;
            __syncthreads();   // the tile's previous readers are done
#pragma unroll
            for (int i = 0; i < 8; ++i)
                *reinterpret_cast<float4 *>(&tile[w][i * 8 + (lane >> 3)][(lane & 7) * 4]) = cur[i];
            if (it + 1 < total) gload(it + 1, nxt);
            __syncthreads();
            for (int c = 0; c < cw; c += 4) {
                const float4 a4 = *reinterpret_cast<const float4 *>(&xq[c0 + c]);
                const float4 c4 = *reinterpret_cast<const float4 *>(&tile[w][lane][c]);
                dot = (c0 + c) ? __fmaf_rn(a4.x, c4.x, dot) : __fmul_rn(a4.x, c4.x);
                dot = __fmaf_rn(a4.y, c4.y, dot);
                dot = __fmaf_rn(a4.z, c4.z, dot);
                dot = __fmaf_rn(a4.w, c4.w, dot);
            }
            const int row = (it / nchunk) * kQueryThreads + w * 64 + lane;
            if (c0 + CH >= D && row < N) push(row, dot);
#pragma unroll
            for (int i = 0; i < 8; ++i) cur[i] = nxt[i];
        }
    } else {
        for (int row = t; row < N; row += kQueryThreads) {
            const float *__restrict__ xc = xb + (size_t)row * D;
            float dot = __fmul_rn(xq[0], xc[0]);
            for (int c = 1; c < D; ++c) dot = __fmaf_rn(xq[c], xc[c], dot);
            push(row, dot);
        }
    }
    __syncthreads();
    const int n = list_n;
    if (n > kQueryList || n < k) {   // (n < k cannot happen for finite inputs; either way the list mode decides)
        if (t == 0) redo[b] = 1;
        return;
    }
    if (t < n) {
        const float v = list_v[t];
        const int j = list_j[t];
        int rank = 0;
        for (int i = 0; i < n; ++i) {
            const float ov = list_v[i];
            const int oj = list_j[i];
            rank += (ov < v || (ov == v && oj < j)) ? 1 : 0;
        }
        if (rank < k) out[((size_t)b * N + q) * k + rank] = (int64_t)j;
    }
}

// pcb_knn through the screening pass + exact recomputation of the uncertified queries.
// workspace (pcb_knn_screen_workspace bytes) = { redo [B] i32 | qcount [B] i32 | qlist [B*N] i32 | qtk [B*N] f32 |
//                                               pad to 16 B | xh [B*N*DP] bf16 | xl [B*N*DP] bf16 }
inline size_t screen_head_bytes(int B, int N) { return ((sizeof(int) * (2 * (size_t)B + 2 * (size_t)B * N)) + 15) / 16 * 16; }

template <int DP>
int launch_knn_screened(const float *x, float *norms, int B, int N, int D, int k, int64_t *out, void *workspace,
                        hipStream_t st)
{
    const long rows = (long)B * N;
    int *redo = (int *)workspace;
    int *qcount = redo + B;
    int *qlist = qcount + B;
    float *qtk = (float *)(qlist + (size_t)B * N);
    unsigned short *xh = (unsigned short *)((char *)workspace + screen_head_bytes(B, N));
    unsigned short *xl = xh + (size_t)rows * DP;
    if (pcb_zero_async(workspace, sizeof(int) * 2 * (size_t)B, st) != PCB_OK) return PCB_ERR_LAUNCH;
    hipLaunchKernelGGL(knn_norms_kernel, dim3((unsigned)((rows + kThreads - 1) / kThreads)), dim3(kThreads), 0, st, x,
                       rows, D, norms);
    hipLaunchKernelGGL(knn_split_kernel, dim3((unsigned)((rows * DP + kThreads - 1) / kThreads)), dim3(kThreads), 0, st, x,
                       rows, D, DP, xh, xl);
    const dim3 grid((N + 127) / 128, B);
    int idx_bits = 6;   // the index field of the packed keys: N <= 65536 (screen_serves)
    while ((1 << idx_bits) < N) ++idx_bits;
    const dim3 sgrid(((grid.x * (unsigned)B + 7) / 8) * 8);   // linear, XCD-aware (see the kernel)
    if (k <= 8)
        hipLaunchKernelGGL((knn_screen_kernel<DP, 8, 12>), sgrid, dim3(kThreads), 0, st, x, (const unsigned short *)xh,
                           (const unsigned short *)xl, (const float *)norms, B, N, D, k, idx_bits, out, qlist, qtk, qcount);
    else
        hipLaunchKernelGGL((knn_screen_kernel<DP, 20, 24>), sgrid, dim3(kThreads), 0, st, x, (const unsigned short *)xh,
                           (const unsigned short *)xl, (const float *)norms, B, N, D, k, idx_bits, out, qlist, qtk, qcount);
    // the uncertified queries: a few per scene -> one workgroup each; many (or a flagged scene) -> the list mode
    hipLaunchKernelGGL(knn_query_kernel, dim3(kPerQueryMax, B), dim3(kQueryThreads), 0, st, x, (const float *)norms, N, D, k,
                       out, (const int *)qlist, (const float *)qtk, (const int *)qcount, redo);
    if (k <= 8)
        hipLaunchKernelGGL((knn_mfma_kernel<DP, 8>), grid, dim3(kThreads), 0, st, x, (const float *)norms, N, D, k, out,
                           (const int *)nullptr, 0, (const int *)qlist, (const int *)qcount, kPerQueryMax,
                           (const int *)redo);
    else
        hipLaunchKernelGGL((knn_mfma_kernel<DP, 20>), grid, dim3(kThreads), 0, st, x, (const float *)norms, N, D, k, out,
                           (const int *)nullptr, 0, (const int *)qlist, (const int *)qcount, kPerQueryMax,
                           (const int *)redo);
    pcb_account(4.0 * (double)D * N * B + 8.0 * (double)N * k * B);
    return pcb_check_launch();
}

inline int screen_dp(int D) { return D <= 32 ? 32 : (D <= 64 ? 64 : 128); }
inline bool screen_serves(int N, int D, int k) { return D >= 32 && D <= 128 && k <= 20 && N >= 64 && N <= 65536; }

}  // namespace

extern "C" long pcb_knn_screen_workspace(int B, int N, int D, int k)
{
    if (B <= 0 || N <= 0 || !screen_serves(N, D, k)) return 0;
    return (long)(screen_head_bytes(B, N) + 4 * (size_t)B * N * screen_dp(D));
}

// pcb_knn for 32 <= D <= 128, k <= 20 through the split-bf16 screening pass: same output, same norms.  Shapes it does
// not serve (pcb_knn_screen_workspace == 0) go to pcb_knn.  workspace[B .. 2B) (int32) holds, after the call, the number of
// queries of every scene that the exact kernel recomputed.
extern "C" int pcb_knn_screened(const float *x, int B, int N, int D, int k, float *norms, void *workspace, int64_t *out_idx,
                                void *stream)
{
    if (!x || !norms || !out_idx || B <= 0 || N <= 0 || D <= 0) return PCB_ERR_INVALID_ARG;
    if (k < 1 || k > 32 || k > N) return PCB_ERR_INVALID_ARG;
    if (D > 128) return PCB_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (!workspace || !screen_serves(N, D, k)) return pcb_knn(x, B, N, D, k, norms, out_idx, stream);
    if (D <= 32) return launch_knn_screened<32>(x, norms, B, N, D, k, out_idx, workspace, st);
    if (D <= 64) return launch_knn_screened<64>(x, norms, B, N, D, k, out_idx, workspace, st);
    return launch_knn_screened<128>(x, norms, B, N, D, k, out_idx, workspace, st);
}

extern "C" int pcb_knn(const float *x, int B, int N, int D, int k, float *norms, int64_t *out_idx, void *stream)
{
    if (!x || !norms || !out_idx || B <= 0 || N <= 0 || D <= 0) return PCB_ERR_INVALID_ARG;
    if (k < 1 || k > 32 || k > N) return PCB_ERR_INVALID_ARG;
    if (D > 128) return PCB_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (D <= 4) return launch_knn<4>(x, norms, B, N, D, k, out_idx, st);
    if (D <= 8) return launch_knn<8>(x, norms, B, N, D, k, out_idx, st);
    if (D <= 16) return launch_knn<16>(x, norms, B, N, D, k, out_idx, st);
    if (D <= 32) return launch_knn<32>(x, norms, B, N, D, k, out_idx, st);
    if (D <= 64) return launch_knn<64>(x, norms, B, N, D, k, out_idx, st);
    return launch_knn<128>(x, norms, B, N, D, k, out_idx, st);
}

// pcb_knn with a per-scene switch: scene b is computed only if only_if[b * stride] != 0 (NULL: all).
int pcb_knn_flagged(const float *x, int B, int N, int D, int k, float *norms, int64_t *out_idx, const int *only_if,
                    int only_if_stride, hipStream_t st)
{
    if (D <= 4) return launch_knn<4>(x, norms, B, N, D, k, out_idx, st, only_if, only_if_stride);
    if (D <= 8) return launch_knn<8>(x, norms, B, N, D, k, out_idx, st, only_if, only_if_stride);
    if (D <= 16) return launch_knn<16>(x, norms, B, N, D, k, out_idx, st, only_if, only_if_stride);
    if (D <= 32) return launch_knn<32>(x, norms, B, N, D, k, out_idx, st, only_if, only_if_stride);
    if (D <= 64) return launch_knn<64>(x, norms, B, N, D, k, out_idx, st, only_if, only_if_stride);
    return launch_knn<128>(x, norms, B, N, D, k, out_idx, st, only_if, only_if_stride);
}
