// Host-side runtime of the shared pointwise MLP stacks (see include/pcb_hip.h):
//
//   pcb_mlp_stack_forward / pcb_mlp_stack_backward   enqueue ALL kernels of a stack of
//       Conv(1x1) -> BatchNorm -> activation layers [-> max over the neighbour axis] -- the loop
//       the reference writes as `for i, conv in enumerate(self.mlp_convs): x = F.relu(bn(conv(x)))`
//       (models/pointnet2_utils.py:149-154, :207-209, :353-356; models/DGCNN.py:134-148) -- from
//       one call, so the host pays one foreign call per stack and direction instead of one per
//       kernel (a PointNet++ MSG step has ~260 such launches; at ~20 us of interpreter time each
//       they would out-last the GPU work).
//   pcb_timer_*   HIP-event timing of the roofline kernel family (the gemm_nt launches) on the
//       stream they are launched on, for bench.py.
//
// No device code here: everything goes through the entry points of gemm.hip / rowbn.hip.
#include <hip/hip_runtime.h>

#include <stdio.h>
#include <stdlib.h>

#include <mutex>
#include <utility>
#include <vector>

#include "gemm_shared.h"

// ---- roofline timer -------------------------------------------------------------------------
// Two instruments for bench.py, both off unless pcb_timer_start armed them:
//   * HIP events around the launches of two kernel categories -- 0: the gemm_nt family (the roofline
//     kernel), 1: farthest point sampling (the step's latency chain) -- on the stream they run on;
//   * a byte counter: every entry point adds the ALGORITHMIC HBM bytes of its launch (operands read
//     once + outputs written once), the whole-step figure of the bench line.
namespace {
constexpr int kCats = 2;
struct Timer {
    std::mutex mu;
    bool armed = false;    // between pcb_timer_start and pcb_timer_stop
    bool on = false;       // event sampling enabled right now
    struct Ev { hipEvent_t first, second, third; int cat; double bytes; };  // start | stop | a second stop right behind it
    std::vector<Ev> events;
    std::vector<hipEvent_t> pool;
    struct Rec { double bytes; int pro; long R; int N, K; };
    std::vector<Rec> each;  // the timed category-0 launches (PCB_TIMER_VERBOSE listing)
    double all_bytes = 0.0;   // every accounted launch since pcb_timer_start
    long all_launches = 0;
    // results of the last pcb_timer_stop, per category
    long n[kCats] = {0, 0};
    double ms[kCats] = {0.0, 0.0}, bytes[kCats] = {0.0, 0.0};
    hipEvent_t get()
    {
        if (!pool.empty()) {
            hipEvent_t e = pool.back();
            pool.pop_back();
            return e;
        }
        hipEvent_t e;
        return hipEventCreate(&e) == hipSuccess ? e : nullptr;
    }
} g_timer;
}  // namespace

// called by every entry point: algorithmic bytes of the launch it enqueued
void pcb_account(double bytes)
{
    if (!g_timer.armed) return;
    std::lock_guard<std::mutex> lk(g_timer.mu);
    g_timer.all_bytes += bytes;
    g_timer.all_launches += 1;
}

// called by the timed entry points around their launch
void pcb_timer_begin_cat(hipStream_t st, hipEvent_t *stop, int cat)
{
    *stop = nullptr;
    if (!g_timer.on) return;
    std::lock_guard<std::mutex> lk(g_timer.mu);
    if (!g_timer.on) return;
    hipEvent_t a = g_timer.get(), b = g_timer.get(), c = g_timer.get();
    if (!a || !b || !c) return;
    (void)hipEventRecord(a, st);
    g_timer.events.push_back({a, b, c, cat, 0.0});
    *stop = b;
}
void pcb_timer_begin(hipStream_t st, hipEvent_t *stop) { pcb_timer_begin_cat(st, stop, 0); }

void pcb_timer_end(hipStream_t st, hipEvent_t stop, double bytes, int pro, long R, int N, int K)
{
    static const bool trace = getenv("PCB_NT_TRACE") != nullptr;  // launch list for tools/nt_bench.py
    if (trace && R > 0) fprintf(stderr, "[pcb_nt] %d %ld %d %d\n", pro, R, N, K);
    pcb_account(bytes);
    if (!stop) return;
    (void)hipEventRecord(stop, st);
    std::lock_guard<std::mutex> lk(g_timer.mu);
    // An event is a packet of its own on the queue: elapsed(start, stop) holds the kernel PLUS the
    // processing of one event packet.  A third event recorded right behind `stop` measures exactly
    // that (nothing lies between them) and is subtracted when the samples are read.
    for (auto it = g_timer.events.rbegin(); it != g_timer.events.rend(); ++it) {
        if (it->second == stop) {
            (void)hipEventRecord(it->third, st);
            it->bytes = bytes;
            if (it->cat == 0) g_timer.each.push_back({bytes, pro, R, N, K});
            break;
        }
    }
}

extern "C" int pcb_timer_start(void)
{
    std::lock_guard<std::mutex> lk(g_timer.mu);
    for (auto &p : g_timer.events) {
        g_timer.pool.push_back(p.first);
        g_timer.pool.push_back(p.second);
        g_timer.pool.push_back(p.third);
    }
    g_timer.events.clear();
    g_timer.each.clear();
    g_timer.all_bytes = 0.0;
    g_timer.all_launches = 0;
    g_timer.armed = true;
    g_timer.on = true;
    return PCB_OK;
}

extern "C" int pcb_timer_enable(int on)
{
    std::lock_guard<std::mutex> lk(g_timer.mu);
    g_timer.on = g_timer.armed && on != 0;
    return PCB_OK;
}

extern "C" int pcb_timer_stop(long *launches, double *milliseconds, double *bytes)
{
    std::lock_guard<std::mutex> lk(g_timer.mu);
    g_timer.on = false;
    g_timer.armed = false;
    for (int c = 0; c < kCats; ++c) {
        g_timer.n[c] = 0;
        g_timer.ms[c] = g_timer.bytes[c] = 0.0;
    }
    const bool verbose = getenv("PCB_TIMER_VERBOSE") != nullptr;
    size_t i = 0;
    for (auto &p : g_timer.events) {
        float t = 0.0f, gap = 0.0f;
        if (hipEventSynchronize(p.third) == hipSuccess && hipEventElapsedTime(&t, p.first, p.second) == hipSuccess &&
            hipEventElapsedTime(&gap, p.second, p.third) == hipSuccess) {
            if (verbose) fprintf(stderr, "[pcb_timer] cat %d raw %.1f us, event packet %.1f us\n", p.cat, t * 1e3, gap * 1e3);
            t = t > gap ? t - gap : 0.0f;
            g_timer.ms[p.cat] += t;
            g_timer.bytes[p.cat] += p.bytes;
            g_timer.n[p.cat] += 1;
            if (verbose && p.cat == 0 && i < g_timer.each.size())
                fprintf(stderr, "[pcb_timer] %zu pro=%d R=%ld N=%d K=%d: %.1f MB in %.1f us = %.2f TB/s\n", i,
                        g_timer.each[i].pro, g_timer.each[i].R, g_timer.each[i].N, g_timer.each[i].K,
                        g_timer.each[i].bytes / 1e6, t * 1e3, g_timer.each[i].bytes / (t * 1e-3) / 1e12);
        }
        if (p.cat == 0) ++i;
        g_timer.pool.push_back(p.first);
        g_timer.pool.push_back(p.second);
        g_timer.pool.push_back(p.third);
    }
    g_timer.events.clear();
    if (launches) *launches = g_timer.n[0];
    if (milliseconds) *milliseconds = g_timer.ms[0];
    if (bytes) *bytes = g_timer.bytes[0];
    return PCB_OK;
}

extern "C" int pcb_timer_read(int category, long *launches, double *milliseconds, double *bytes)
{
    std::lock_guard<std::mutex> lk(g_timer.mu);
    if (category == -1) {  // every accounted launch between start and stop (not event-timed)
        if (launches) *launches = g_timer.all_launches;
        if (milliseconds) *milliseconds = 0.0;
        if (bytes) *bytes = g_timer.all_bytes;
        return PCB_OK;
    }
    if (category < 0 || category >= kCats) return PCB_ERR_INVALID_ARG;
    if (launches) *launches = g_timer.n[category];
    if (milliseconds) *milliseconds = g_timer.ms[category];
    if (bytes) *bytes = g_timer.bytes[category];
    return PCB_OK;
}

// ---- stack descriptors ------------------------------------------------------------------------
namespace {
constexpr int kSlots = PCB_STACK_DESC_SLOTS;  // int64 slots per layer, see include/pcb_hip.h
enum { S_W = 0, S_BIAS, S_GAMMA, S_BETA, S_RMEAN, S_RVAR, S_C, S_K, S_TRAIN, S_Y, S_DW, S_DGAMMA, S_DBETA, S_DBIAS, S_NBT, S_EXT,
       S_CENTRE, S_CFLAGS };
static_assert(kSlots == 18, "descriptor layout");
// S_CENTRE / S_CFLAGS (bf16 rows, forward only): the layer's y is stored centred (gemm.hip RedArgs::centre).
//   S_CENTRE  fp32 [C] owned by the caller across calls (one per BatchNorm layer) or 0: the centre of a TRAINING-mode layer,
//             read by its GEMM, moved to this batch's mean by its finalize kernel;
//   S_CFLAGS  bit 0 PROBE: the buffer holds no estimate yet (zeros): the layer runs GEMM + finalize once to learn the batch
//             mean and once more centred on it;  bit 1: centre an EVAL-mode layer on running_mean - bias (kept in row 0 of
//             the layer's stz block; its finalize kernel then runs BEFORE its GEMM).
enum { CF_PROBE = 1, CF_EVAL = 2 };
// S_EXT (layer 0 of a plain bf16 stack only): host address of 7 int64 or 0 --
//   [0] add1, [1] sh1, [2] add2 (or 0), [3] sh2: fp32 rows [R >> sh, C] added to the layer's product before rounding, each
//       standing for 2^sh consecutive rows (pcb_gemm_nt_stats_add_bf16);  [4] d add1, [5] d add2: their gradients (backward);
//   [6] row stride of the layer's weight in floats (0 = k): a column slice of a wider weight read in place.
enum { X_ADD1 = 0, X_SH1, X_ADD2, X_SH2, X_DADD1, X_DADD2, X_LDW };

struct Layer {
    const float *w, *bias, *gamma, *beta;
    float *rmean, *rvar;
    int C, k, training;
    void *y;
    float *dW, *dgamma, *dbeta, *dbias;
    long long *nbt;  // num_batches_tracked or NULL
    const long long *ext;  // S_EXT or NULL
    float *centre;   // S_CENTRE or NULL
    int cflags;      // S_CFLAGS
    int kp;          // padded input width (= previous layer's C, or the stack's Kp)
    long wp_off;     // element offsets into the weight buffer
    long wt_off;     // -1: no transposed copy
    long st_off;     // float offset of this layer's [10][C] constants
};

// The kernels of one arithmetic mode (PCB_DTYPE_BF16: gemm.hip + rowbn.hip *_bf16; PCB_DTYPE_F32:
// gemm_f32.hip + rowbn.hip *_f32).  The runtime below is the same for both: only the storage type
// of the rows differs.
struct Ops {
    int quantum;      // columns per 16-byte chunk: widths are multiples of it
    size_t elem;      // bytes per row element
    decltype(&pcb_prep_weights_zero_bf16) prep_zero;
    decltype(&pcb_gemm_nt_bf16) gemm_nt;
    decltype(&pcb_gemm_nt_red_bf16) gemm_nt_red;
    decltype(&pcb_gemm_tn_bf16) gemm_tn;
    decltype(&pcb_bn_act_bf16) bn_act;
    decltype(&pcb_bn_act_max_bf16) bn_act_max;
    decltype(&pcb_bn_act_bwd_reduce_bf16) bwd_reduce;
    decltype(&pcb_bn_act_max_bwd_reduce_bf16) max_bwd_reduce;
};
const Ops kOps[2] = {
    {8, 2, pcb_prep_weights_zero_bf16, pcb_gemm_nt_bf16, pcb_gemm_nt_red_bf16, pcb_gemm_tn_bf16, pcb_bn_act_bf16,
     pcb_bn_act_max_bf16, pcb_bn_act_bwd_reduce_bf16, pcb_bn_act_max_bwd_reduce_bf16},
    {4, 4, pcb_prep_weights_zero_f32, pcb_gemm_nt_f32, pcb_gemm_nt_red_f32, pcb_gemm_tn_f32, pcb_bn_act_f32,
     pcb_bn_act_max_f32, pcb_bn_act_bwd_reduce_f32, pcb_bn_act_max_bwd_reduce_f32},
};

template <typename T>
T *ptr(long long v) { return reinterpret_cast<T *>(static_cast<uintptr_t>(v)); }

int parse(int L, const long long *desc, int Kp, int need_wt0, bool gathered, int quantum, Layer *out)
{
    if (L < 1 || L > PCB_STACK_MAX_LAYERS || !desc) return PCB_ERR_INVALID_ARG;
    long woff = 0, soff = 0;
    int kp = Kp;
    for (int l = 0; l < L; ++l) {
        const long long *d = desc + (long)kSlots * l;
        Layer &a = out[l];
        a.w = ptr<const float>(d[S_W]);
        a.bias = ptr<const float>(d[S_BIAS]);
        a.gamma = ptr<const float>(d[S_GAMMA]);
        a.beta = ptr<const float>(d[S_BETA]);
        a.rmean = ptr<float>(d[S_RMEAN]);
        a.rvar = ptr<float>(d[S_RVAR]);
        a.C = (int)d[S_C];
        a.k = (int)d[S_K];
        a.training = (int)d[S_TRAIN];
        a.y = ptr<void>(d[S_Y]);
        a.dW = ptr<float>(d[S_DW]);
        a.dgamma = ptr<float>(d[S_DGAMMA]);
        a.dbeta = ptr<float>(d[S_DBETA]);
        a.dbias = ptr<float>(d[S_DBIAS]);
        a.nbt = ptr<long long>(d[S_NBT]);
        a.ext = ptr<const long long>(d[S_EXT]);
        a.centre = ptr<float>(d[S_CENTRE]);
        a.cflags = (int)d[S_CFLAGS];
        if (a.ext && (l > 0 || gathered)) return PCB_ERR_INVALID_ARG;
        if (gathered && l == 0) {
            // layer 0 = gather_add of per-point products: no weights of its own in this call
            if (!a.y || a.C <= 0 || (a.C % quantum)) return PCB_ERR_INVALID_ARG;
            a.kp = 0;
            a.wp_off = woff;
            a.wt_off = -1;
            a.st_off = soff;
            soff += 10L * a.C;
            kp = a.C;
            continue;
        }
        if (!a.w || !a.y || a.C <= 0 || (a.C % quantum) || a.k <= 0 || a.k > kp || (kp % quantum)) return PCB_ERR_INVALID_ARG;
        a.kp = kp;
        a.wp_off = woff;
        woff += (long)a.C * kp;
        if (l > 0 || need_wt0) {
            a.wt_off = woff;
            woff += (long)a.C * kp;
        } else {
            a.wt_off = -1;
        }
        a.st_off = soff;
        soff += 10L * a.C;
        kp = a.C;
    }
    return PCB_OK;
}

inline float *row(float *stz, const Layer &a, int r) { return stz + a.st_off + (long)r * a.C; }

// A second stream for the weight-gradient GEMMs of a stack's backward pass: dW is needed by nobody
// before the optimizer, so gemm_tn(l) runs beside gemm_nt(l) (the input gradient, which IS on the
// critical path).  Both are bandwidth kernels that leave latency bubbles when alone on the chip.
// Fork/join with events; the join at the end of every backward call keeps the caller's
// stream-ordered view of all buffers intact.  Opt-in (PCB_WGRAD_STREAM=1): measured -2 % step time
// on PN2-MSG (9.92 -> 9.72 ms under hipGraph replay), but every gemm_nt then shares HBM with a
// gemm_tn, which makes per-kernel durations (and bench.py's roofline figure) meaningless.
struct SideStream {
    hipStream_t st = nullptr;
    hipEvent_t ring[64] = {};
    unsigned next = 0;
    bool failed = false;
    hipEvent_t event()
    {
        hipEvent_t &e = ring[next++ & 63];
        if (!e && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) e = nullptr;
        return e;
    }
};
SideStream *side_stream()
{
    static const bool enabled = getenv("PCB_WGRAD_STREAM") && atoi(getenv("PCB_WGRAD_STREAM")) != 0;
    if (!enabled) return nullptr;
    static SideStream per_device[16];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    SideStream &s = per_device[dev];
    if (!s.st && !s.failed && hipStreamCreateWithFlags(&s.st, hipStreamNonBlocking) != hipSuccess) s.failed = true;
    return s.failed ? nullptr : &s;
}

// Slabs a gemm_nt launch of this stack gets: the library's preference for the concurrency hint read
// at the top of the call, capped by what the caller's `parts` buffer holds.  The same number goes to
// the GEMM (its grid) and to the finalize kernel that adds the slabs.
inline int slabs_for(int pro, long R, int C, int busy, int parts_slabs)
{
    const long want = pcb_nt_grid_x(pro, R, C, busy);
    return (int)(want < parts_slabs ? want : parts_slabs);
}
}  // namespace

#define PCB_TRY(expr)                \
    do {                             \
        const int s_ = (expr);       \
        if (s_ < 0) return s_;       \
    } while (0)

extern "C" long pcb_mlp_stack_wbuf_elems(int L, const long long *desc, int Kp, int need_wt0)
{
    if (L < 1 || L > PCB_STACK_MAX_LAYERS || !desc) return 0;
    long total = 0;
    int kp = Kp;  // Kp == 0: layer 0 is a gathered layer and holds no weights here
    for (int l = 0; l < L; ++l) {
        const int C = (int)desc[(long)kSlots * l + S_C];
        total += (long)C * kp * ((l > 0 || need_wt0) ? 2 : 1);
        kp = C;
    }
    return total;
}

namespace {
// Layers whose dy is written out once (pcb_dy_rows_bf16) instead of being rebuilt in the prologues of their
// gradient GEMMs: bf16 rows, dense dz, wider than the A-resident input-gradient kernel takes (C > 256) and
// more than 256 inputs -- both GEMMs then walk >= 3 column tiles of the partner matrix.
inline bool materialise_dy(int dtype, bool pooled, int C, int kp) { return dtype == PCB_DTYPE_BF16 && !pooled && C > 256 && C <= 2048 && kp > 256; }

// row stride of the two gradient ping-pong slots: the widest dz they hold -- and the top layer's dy where that is
// written out (slot L & 1 is free while layer L-1, which reads the caller's g, runs)
inline int dz_stride(int dtype, int L, const Layer *ly, int Kp, int pool, bool gathered)
{
    int maxw = Kp > 8 ? Kp : 8;
    for (int l = 0; l + 1 < L; ++l) maxw = ly[l].C > maxw ? ly[l].C : maxw;
    const Layer &top = ly[L - 1];
    if (!(gathered && L == 1) && materialise_dy(dtype, pool != 0, top.C, top.kp) && top.C > maxw) maxw = top.C;
    return maxw;
}
}  // namespace

extern "C" long pcb_mlp_stack_dzbuf_elems(int dtype, int L, const long long *desc, long R, int Kp, int pool, int gathered)
{
    Layer ly[PCB_STACK_MAX_LAYERS];
    if (dtype != PCB_DTYPE_BF16 && dtype != PCB_DTYPE_F32) return 0;
    if (parse(L, desc, gathered ? 0 : Kp, 1, gathered != 0, kOps[dtype].quantum, ly) != PCB_OK || R <= 0) return 0;
    const Layer &top = ly[L - 1];
    const bool top_mat = !(gathered && L == 1) && materialise_dy(dtype, pool != 0, top.C, top.kp);
    if (L == 1 && !top_mat) return 0;
    return 2L * R * dz_stride(dtype, L, ly, gathered ? 0 : Kp, pool, gathered != 0);
}

namespace {
struct Gather {
    float *u, *v;  // forward: u, v (v optional);  backward: du, dv
    const int64_t *idx;
    int B, N, S, ns;
    const float *xyz, *ctr;  // optional coordinate term Wx (x_j - c_s)
    float *wx;               // forward: Wx [C,3] with row stride ldw;  backward: dWx [C,3]
    int ldw;
    int det;                        // backward: reproducible mode (no atomics; segsum.hip)
    const int *order;               // det: inverted index of idx (ops.det_index): grouped rows by source point
    const long long *offsets;       //      [B*N + 1]
};
bool parse_gather(const long long *g, Gather *out)
{
    if (!g) return false;
    out->u = ptr<float>(g[0]);
    out->v = ptr<float>(g[1]);
    out->idx = ptr<const int64_t>(g[2]);
    out->B = (int)g[3];
    out->N = (int)g[4];
    out->S = (int)g[5];
    out->ns = (int)g[6];
    out->xyz = ptr<const float>(g[7]);
    out->ctr = ptr<const float>(g[8]);
    out->wx = ptr<float>(g[9]);
    out->ldw = (int)g[10];
    out->det = (int)(g[11] & 1);
    out->order = ptr<const int>(g[12]);
    out->offsets = ptr<const long long>(g[13]);
    return true;
}
}  // namespace

extern "C" int pcb_mlp_stack_forward(int dtype, int L, const long long *desc, const double *fdesc, const void *x,
                                     long R, int Kp, int perm, int act, int pool, int need_wt0, int stat_repeat,
                                     const long long *gather, void *wbuf, float *stz, float *parts, int parts_slabs,
                                     const pcb_sync *sync, void *out, unsigned char *argmax, void *stream)
{
    if (dtype != PCB_DTYPE_BF16 && dtype != PCB_DTYPE_F32) return PCB_ERR_INVALID_ARG;
    const Ops &op = kOps[dtype];
    Layer ly[PCB_STACK_MAX_LAYERS];
    Gather ga;
    const bool gathered = parse_gather(gather, &ga);
    if (gathered) {
        if (dtype != PCB_DTYPE_BF16) return PCB_ERR_UNSUPPORTED;  // the gathered first layer exists for bf16 rows
        if (!ga.u || !ga.idx || (long)ga.B * ga.S * ga.ns != R || !parts) return PCB_ERR_INVALID_ARG;
        Kp = 0;
    }
    // need_wt0 bit 1: wbuf and stz still hold the operands and BatchNorm constants an earlier
    // eval-mode call prepared from the same, unchanged parameters -- no preparation, no finalize
    const bool ready = (need_wt0 & 2) != 0;
    // bit 2: wbuf holds the operands of the current weights already (one table launch per optimiser step prepared
    // every stack's): no preparation launch here
    const bool wready = (need_wt0 & 4) != 0;
    need_wt0 &= 1;
    PCB_TRY(parse(L, desc, Kp, need_wt0, gathered, op.quantum, ly));
    if (!fdesc || (!x && !gathered) || (!wbuf && !(gathered && L == 1)) || !stz || !out || R <= 0 || pool < 0 ||
        (pool && (!argmax || R % pool)))
        return PCB_ERR_INVALID_ARG;
    if (parts && parts_slabs < 1) return PCB_ERR_INVALID_ARG;
    if (sync && (!sync->allreduce || sync->global_rows < R || !parts)) return PCB_ERR_INVALID_ARG;
    if (ready)
        for (int l = 0; l < L; ++l)
            if (ly[l].training) return PCB_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    char *wb = (char *)wbuf;
    const int busy = pcb_busy_cus();  // one reading of the hint for the whole call

    long stz_floats = 0;
    for (int l = 0; l < L; ++l) stz_floats += 10L * ly[l].C;
    // stz is cleared by the first weight-preparation launch (or a zero launch if there is none).  What needs clearing is
    // the single accumulation slab of a backward reduction without slabs (rows 6, 7); with a slab buffer every row of
    // stz is written before it is read, so a call with prepared weights and `parts` clears nothing.
    bool cleared = ready || (wready && parts && parts_slabs > 1);

    // GEMM operands of all layers (chunks of 8 layers per launch)
    for (int l0 = gathered ? 1 : 0; l0 < L && !ready && !wready; l0 += 8) {
        const int n = L - l0 < 8 ? L - l0 : 8;
        long long pd[8 * 8];
        for (int i = 0; i < n; ++i) {
            const Layer &a = ly[l0 + i];
            long long *d = pd + 8 * i;
            d[0] = (long long)(uintptr_t)a.w;
            d[1] = (long long)(uintptr_t)(wb + a.wp_off * op.elem);
            d[2] = a.wt_off >= 0 ? (long long)(uintptr_t)(wb + a.wt_off * op.elem) : 0;
            d[3] = a.C;
            d[4] = a.k;
            d[5] = a.kp;
            d[6] = (l0 + i == 0) ? perm : 0;
            d[7] = a.ext ? a.ext[X_LDW] : 0;
        }
        PCB_TRY(op.prep_zero(n, pd, cleared ? nullptr : stz, cleared ? 0 : stz_floats, stream));
        cleared = true;
    }
    if (!cleared && pcb_zero_async(stz, stz_floats * sizeof(float), st) != PCB_OK) return PCB_ERR_LAUNCH;

    const void *cur = x;
    for (int l = 0; l < L; ++l) {
        const Layer &a = ly[l];
        const bool stats = a.training != 0;
        if (stats && !parts) return PCB_ERR_INVALID_ARG;
        const float *pscale = l ? row(stz, ly[l - 1], 2) : nullptr;
        const float *pshift = l ? row(stz, ly[l - 1], 3) : nullptr;
        // rows stored centred (bf16 rows only): a training-mode layer on the caller's persistent estimate of its batch mean,
        // an eval-mode layer on running_mean - bias (row 0 of its constants, written by its finalize kernel first)
        const bool bf16 = dtype == PCB_DTYPE_BF16;
        float *centre = nullptr;
        if (bf16 && stats && a.centre) centre = a.centre;
        const bool eval_centre = bf16 && !stats && (a.cflags & CF_EVAL) && a.rmean && a.rvar;
        if (eval_centre) centre = row(stz, a, 0);
        int nparts = 0;
        auto finalize = [&](int cmode) -> int {
            const float *sums = stats ? parts : nullptr;
            long rows = R;
            int np = nparts;
            if (stats && sync) {
                // SyncBatchNorm: the local totals (2C floats) travel through the caller's all-reduce;
                // the finalize kernel then sees the statistics of all ranks' rows
                float *tot = row(stz, a, 0);
                PCB_TRY(pcb_sum_slabs(parts, nparts, 2 * a.C, tot, stream));
                if (sync->allreduce(tot, 2 * a.C, sync->ctx) != 0) return PCB_ERR_LAUNCH;
                sums = tot;
                np = 1;
                rows = sync->global_rows;
            }
            return pcb_bn_finalize_centred(sums, np, rows, rows * (stat_repeat > 1 ? stat_repeat : 1), a.C, a.gamma, a.beta,
                                           a.bias, a.rmean, a.rvar, (float)fdesc[2 * l], (float)fdesc[2 * l + 1], a.training,
                                           row(stz, a, 2), row(stz, a, 3), row(stz, a, 4), row(stz, a, 5), a.nbt, centre,
                                           cmode, stream);
        };
        auto produce = [&]() -> int {
            if (gathered && l == 0) {
                const int want = pcb_gather_add_partials(R, a.C);
                nparts = want < parts_slabs ? want : parts_slabs;
                return pcb_gather_add_bf16(ga.u, ga.v, ga.idx, ga.B, ga.N, ga.S, ga.ns, a.C, ga.xyz, ga.ctr, ga.wx, ga.ldw, a.y,
                                           parts, nparts, centre, stream);
            }
            if (a.ext && a.ext[X_ADD1]) {
                // the coarse levels of a repeated concatenation arrive as addends of the product (see S_EXT); the statistics
                // epilogue runs in eval mode too (its slabs are then ignored)
                if (!bf16) return PCB_ERR_UNSUPPORTED;
                if (!parts) return PCB_ERR_INVALID_ARG;
                nparts = slabs_for(0, R, a.C, busy, parts_slabs);
                return pcb_gemm_nt_stats_add_bf16(cur, wb + a.wp_off * op.elem, R, a.C, a.kp, a.y, parts, nparts,
                                                  ptr<const float>(a.ext[X_ADD1]), (int)a.ext[X_SH1],
                                                  ptr<const float>(a.ext[X_ADD2]), (int)a.ext[X_SH2], centre, stream);
            }
            if (centre) {
                // (an eval-mode layer has no use for the slabs, but the centred epilogue is the statistics epilogue)
                if (!parts) return PCB_ERR_INVALID_ARG;
                nparts = slabs_for(l ? 1 : 0, R, a.C, busy, parts_slabs);
                return pcb_gemm_nt_stats_bf16(l ? 1 : 0, cur, pscale, pshift, act, wb + a.wp_off * op.elem, R, a.C, a.kp, a.y,
                                              parts, nparts, centre, stream);
            }
            nparts = stats ? slabs_for(l ? 1 : 0, R, a.C, busy, parts_slabs) : 0;
            return op.gemm_nt(l ? 1 : 0, cur, nullptr, pscale, pshift, nullptr, nullptr, nullptr, nullptr, 0, act,
                              wb + a.wp_off * op.elem, R, a.C, a.kp, a.y, stats ? parts : nullptr, nparts, stream);
        };
        if (eval_centre) {
            // eval mode: the constants do not depend on the rows -- finalize first (it writes the centre), then the GEMM
            if (!ready) PCB_TRY(finalize(1));
            PCB_TRY(produce());
        } else {
            if (centre && (a.cflags & CF_PROBE)) {
                PCB_TRY(produce());
                PCB_TRY(finalize(2));
            }
            PCB_TRY(produce());
            if (!ready) PCB_TRY(finalize(centre ? 1 : 0));
        }
        cur = a.y;
    }
    const Layer &last = ly[L - 1];
    if (pool)
        PCB_TRY(op.bn_act_max(cur, row(stz, last, 2), row(stz, last, 3), R / pool, pool, last.C, act, out, argmax, stream));
    else
        PCB_TRY(op.bn_act(cur, row(stz, last, 2), row(stz, last, 3), R, last.C, act, out, stream));
    return PCB_OK;
}

extern "C" int pcb_mlp_stack_backward(int dtype, int L, const long long *desc, const void *x, const void *g,
                                      const unsigned char *argmax, long R, int Kp, int perm, int act, int pool,
                                      int need_wt0, const long long *gather, const void *wbuf, float *stz,
                                      float *parts, int parts_slabs, const pcb_sync *sync, float *workspace,
                                      void *dzbuf, void *dx, void *stream)
{
    if (dtype != PCB_DTYPE_BF16 && dtype != PCB_DTYPE_F32) return PCB_ERR_INVALID_ARG;
    const Ops &op = kOps[dtype];
    Layer ly[PCB_STACK_MAX_LAYERS];
    Gather ga;
    const bool gathered = parse_gather(gather, &ga);
    if (gathered) {
        if (dtype != PCB_DTYPE_BF16) return PCB_ERR_UNSUPPORTED;
        if (!ga.u || !ga.idx || (long)ga.B * ga.S * ga.ns != R || dx) return PCB_ERR_INVALID_ARG;
        Kp = 0;
    }
    PCB_TRY(parse(L, desc, Kp, need_wt0, gathered, op.quantum, ly));
    if ((!x && !gathered) || !g || (!wbuf && !(gathered && L == 1)) || !stz || (!workspace && !(gathered && L == 1)) ||
        R <= 0 || (pool && !argmax))
        return PCB_ERR_INVALID_ARG;
    if (dx && !need_wt0) return PCB_ERR_INVALID_ARG;
    if (parts && parts_slabs < 1) return PCB_ERR_INVALID_ARG;
    if (sync && (!sync->allreduce || sync->global_rows < R || !parts)) return PCB_ERR_INVALID_ARG;
    const char *wb = (const char *)wbuf;
    const int maxw = dz_stride(dtype, L, ly, Kp, pool, gathered);
    if (L > 1 && !dzbuf) return PCB_ERR_INVALID_ARG;
    const int busy = pcb_busy_cus();  // one reading of the hint for the whole call

    hipStream_t main_st = (hipStream_t)stream;
    SideStream *side = side_stream();
    hipEvent_t tn_done = nullptr;  // completion of the most recent gemm_tn on the side stream
    const void *dz = pool ? nullptr : g;                 // dense upstream gradient (rows)
    const float *dout = pool ? (const float *)g : nullptr;  // pooled upstream gradient (fp32)
    bool have_parts = false;  // sums of layer l already accumulated by the dgrad GEMM of layer l+1
    int have_nparts = 0;
    // every layer's weight-gradient GEMM has its own slab region, so that the slab sums of the whole
    // stack can run as one launch at the end (nothing in this pass reads dW)
    long ws_off[PCB_STACK_MAX_LAYERS];
    {
        long off = 0;
        for (int l = 0; l < L; ++l) {
            ws_off[l] = off;
            if (ly[l].dW && !(gathered && l == 0)) off += pcb_gemm_tn_workspace(R, ly[l].C, ly[l].kp);
        }
    }
    if (!side) pcb_defer_reduces_begin();
    auto layers = [&]() -> int {
        for (int l = L - 1; l >= 0; --l) {
            const Layer &a = ly[l];
            const bool pooled = pool && l == L - 1;
            float *scale = row(stz, a, 2), *shift = row(stz, a, 3), *mean = row(stz, a, 4), *invstd = row(stz, a, 5);
            float *bsums = row(stz, a, 6), *p = row(stz, a, 8), *q = row(stz, a, 9);
            float *sums = bsums;
            int nparts = 1;
            if (have_parts) {
                sums = parts;
                nparts = have_nparts;
            } else if (parts && parts_slabs > 1) {
                // the stack's top layer (nobody above it computed its sums): one slab per workgroup of the
                // reduction pass, summed in slab order by the finalize kernel like the RED epilogue's
                const long units = pooled ? R / pool : R;
                const long lanes = 256 / (a.C / op.quantum) > 0 ? 256 / (a.C / op.quantum) : 1;
                long want = (units + lanes * 4 - 1) / (lanes * 4);
                if (want > 768) want = 768;
                if (want > parts_slabs) want = parts_slabs;
                nparts = want < 2 ? 2 : (int)want;
                sums = parts;
                if (pooled)
                    PCB_TRY(op.max_bwd_reduce(dout, argmax, a.y, scale, shift, mean, invstd, R / pool, pool, a.C, act, parts,
                                              nparts, stream));
                else
                    PCB_TRY(op.bwd_reduce(dz, a.y, scale, shift, mean, invstd, R, a.C, act, parts, nparts, stream));
            } else if (pooled) {
                PCB_TRY(op.max_bwd_reduce(dout, argmax, a.y, scale, shift, mean, invstd, R / pool, pool, a.C, act, bsums,
                                          1, stream));
            } else {
                PCB_TRY(op.bwd_reduce(dz, a.y, scale, shift, mean, invstd, R, a.C, act, bsums, 1, stream));
            }
            const float *gsums = nullptr;
            long rows = R;
            if (sync && a.training) {
                // SyncBatchNorm: p, q from the sums over all ranks' rows; dgamma / dbeta / dbias stay
                // local (the gradient all-reduce averages them like every other parameter gradient)
                float *tot = row(stz, a, 0);
                PCB_TRY(pcb_sum_slabs(sums, nparts, 2 * a.C, tot, stream));
                if (pcb_copy_async(parts, tot, sizeof(float) * 2 * a.C, main_st) != PCB_OK)
                    return PCB_ERR_LAUNCH;
                if (sync->allreduce(parts, 2 * a.C, sync->ctx) != 0) return PCB_ERR_LAUNCH;
                if (sums == bsums && pcb_zero_async(bsums, sizeof(float) * 2 * a.C, main_st) != PCB_OK)
                    return PCB_ERR_LAUNCH;  // what the finalize kernel does for a single atomically accumulated slab
                sums = tot;
                nparts = 1;
                gsums = parts;
                rows = sync->global_rows;
            }
            PCB_TRY(pcb_bn_bwd_finalize(sums, nparts, rows, a.C, scale, mean, invstd, a.training, p, q, a.dgamma, a.dbeta,
                                        a.dbias, gsums, stream));
            have_parts = false;
            int apro = pooled ? 3 : 2;
            const int ns = pooled ? pool : 1;
            if (dzbuf && !(gathered && l == 0) && materialise_dy(dtype, pooled, a.C, a.kp) && (a.dW || l > 0 || dx)) {
                // dy once, as rows: in place over this layer's dz where that is one of the ping-pong slots, in the
                // free slot for the top layer (whose dz is the caller's tensor)
                void *dyb = (l == L - 1) ? (void *)((char *)dzbuf + (size_t)(L & 1) * R * maxw * op.elem) : const_cast<void *>(dz);
                PCB_TRY(pcb_dy_rows_bf16(dz, a.y, scale, shift, p, q, act, R, a.C, dyb, stream));
                dz = dyb;
                apro = 0;
            }
            if (gathered && l == 0) {
                // gathered layer: its input gradients are du (per source point) and dv (per centroid)
                if (pooled && pool != ga.ns) return PCB_ERR_INVALID_ARG;
                if (ga.det && (!ga.order || !ga.offsets)) return PCB_ERR_INVALID_ARG;
                const size_t wx_floats = (size_t)3 * a.C * pcb_scatter_dy_slabs(ga.B, ga.S, a.C, ga.det);
                if (ga.det) {
                    // reproducible mode: du by a segment sum over the inverted index (every row of du is written: no
                    // clearing), dv / dWx by the row-order pass without atomics
                    if (ga.wx && pcb_zero_async(ga.wx, sizeof(float) * wx_floats, main_st) != PCB_OK) return PCB_ERR_LAUNCH;
                    PCB_TRY(pcb_scatter_dy_bf16(pooled ? 1 : 0, dz, a.y, scale, shift, p, q, dout, argmax, act, ga.idx, ga.B,
                                                ga.N, ga.S, ga.ns, a.C, ga.xyz, ga.ctr, nullptr, ga.v, ga.wx, 1, stream));
                    PCB_TRY(pcb_scatter_dy_csr_bf16(pooled ? 1 : 0, dz, a.y, scale, shift, p, q, dout, argmax, act, ga.ns, a.C,
                                                    ga.order, ga.offsets, (long)ga.B * ga.N, ga.u, stream));
                    return PCB_OK;
                }
                if (pcb_zero2_async(ga.u, sizeof(float) * (size_t)ga.B * ga.N * a.C, ga.wx, sizeof(float) * wx_floats,
                                    main_st) != PCB_OK)
                    return PCB_ERR_LAUNCH;
                PCB_TRY(pcb_scatter_dy_bf16(pooled ? 1 : 0, dz, a.y, scale, shift, p, q, dout, argmax, act, ga.idx,
                                            ga.B, ga.N, ga.S, ga.ns, a.C, ga.xyz, ga.ctr, ga.u, ga.v, ga.wx, 0, stream));
                return PCB_OK;
            }
            if (a.ext && a.ext[X_DADD1]) {
                // gradients of the repeated addends: sums of dy over the rows each coarse row stood for
                if (apro != 2 || dtype != PCB_DTYPE_BF16) return PCB_ERR_UNSUPPORTED;
                PCB_TRY(pcb_dy_repeat_sums_bf16(dz, a.y, scale, shift, p, q, act, R, a.C, (int)a.ext[X_SH1],
                                                ptr<float>(a.ext[X_DADD1]), (int)a.ext[X_SH2], ptr<float>(a.ext[X_DADD2]),
                                                stream));
            }
            // a narrow inner layer (C <= 256, K <= 128): input gradient, weight gradient and the sums of the layer below from ONE
            // pass over (dz, y) and the rows below (bwd_fused_kernel) instead of gemm_tn + gemm_nt_red
            static const bool fused_off = getenv("PCB_BWD_FUSED") && atoi(getenv("PCB_BWD_FUSED")) == 0;
            if (!fused_off && dtype == PCB_DTYPE_BF16 && l > 0 && a.dW && parts && !side && apro >= 2 && !a.ext &&
                pcb_bwd_fused_supported(a.C, a.kp)) {
                const Layer &b = ly[l - 1];
                long rps;
                long grid = (R + 63) / 64;                                    // row tiles of the kernel
                // two workgroups per CU it may use (C <= 128), one of eight waves (C <= 256: 131 KB of LDS)
                const long per_cu = a.C > 128 ? 1 : 2;
                const long cap = per_cu * (256 - busy) > 64 ? per_cu * (256 - busy) : 64;
                const long slabs = pcb_tn_splits(R, a.C, a.kp, &rps, 512);     // what the caller's workspace region holds
                grid = grid < cap ? grid : cap;
                grid = grid < slabs ? grid : slabs;
                grid = grid < parts_slabs ? grid : parts_slabs;
                void *dprev = (void *)((char *)dzbuf + (size_t)(l & 1) * R * maxw * op.elem);
                PCB_TRY(pcb_bwd_fused_bf16(apro, dz, a.y, scale, shift, p, q, dout, argmax, ns, act, wb + a.wt_off * op.elem, b.y,
                                           row(stz, b, 2), row(stz, b, 3), row(stz, b, 4), row(stz, b, 5), act, R, a.C, a.kp,
                                           dprev, parts, (int)grid, workspace + ws_off[l], a.dW, a.k, 0, stream));
                have_parts = true;
                have_nparts = (int)grid;
                dz = dprev;
                dout = nullptr;
                continue;
            }
            // weight gradient, in the parameter's own layout -- on the side stream
            hipEvent_t tn_prev = tn_done;  // gemm_tn(l+1): still reading the buffer gemm_nt(l) will write
            if (a.dW) {
                void *tn_stream = stream;
                hipEvent_t fork = side ? side->event() : nullptr;
                if (fork && hipEventRecord(fork, main_st) == hipSuccess &&
                    hipStreamWaitEvent(side->st, fork, 0) == hipSuccess)
                    tn_stream = side->st;
                if (l) {
                    const Layer &b = ly[l - 1];
                    PCB_TRY(op.gemm_tn(apro, dz, a.y, scale, shift, p, q, dout, argmax, ns, act, 1, b.y, row(stz, b, 2),
                                       row(stz, b, 3), act, R, a.C, a.kp, workspace + ws_off[l], a.dW, a.k, 0, tn_stream));
                } else {
                    PCB_TRY(op.gemm_tn(apro, dz, a.y, scale, shift, p, q, dout, argmax, ns, act, 0, x, nullptr, nullptr, 0,
                                       R, a.C, a.kp, workspace + ws_off[l], a.dW, a.k, perm, tn_stream));
                }
                if (tn_stream != stream) {
                    tn_done = side->event();
                    if (!tn_done || hipEventRecord(tn_done, side->st) != hipSuccess) return PCB_ERR_LAUNCH;
                }
            }
            // input gradient
            if (l == 0 && !dx) return PCB_OK;
            if (tn_prev && hipStreamWaitEvent(main_st, tn_prev, 0) != hipSuccess) return PCB_ERR_LAUNCH;
            void *dprev = l ? (void *)((char *)dzbuf + (size_t)(l & 1) * R * maxw * op.elem) : dx;
            const void *wt = wb + a.wt_off * op.elem;
            if (l && a.kp <= 128 && parts) {
                const Layer &b = ly[l - 1];
                have_nparts = slabs_for(apro, R, a.kp, busy, parts_slabs);
                PCB_TRY(op.gemm_nt_red(apro, dz, a.y, scale, shift, p, q, dout, argmax, ns, act, wt, R, a.kp, a.C, dprev,
                                       b.y, row(stz, b, 2), row(stz, b, 3), row(stz, b, 4), row(stz, b, 5), act, parts,
                                       have_nparts, stream));
                have_parts = true;
            } else {
                PCB_TRY(op.gemm_nt(apro, dz, a.y, scale, shift, p, q, dout, argmax, ns, act, wt, R, a.kp, a.C, dprev,
                                   nullptr, 0, stream));
            }
            dz = dprev;
            dout = nullptr;
        }
        return PCB_OK;
    };
    const int status = layers();
    const int flushed = pcb_defer_reduces_flush(main_st);  // the parked slab sums, one launch (also on an error path)
    // join: everything the side stream did is ordered before whatever the caller enqueues next
    if (tn_done && hipStreamWaitEvent(main_st, tn_done, 0) != hipSuccess) return PCB_ERR_LAUNCH;
    return status != PCB_OK ? status : flushed;
}
