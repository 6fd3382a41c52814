// Host-side pieces shared by the bf16 and fp32 row GEMMs (gemm.hip, gemm_f32.hip) and the stack
// runtime (stack.hip): grid / slab sizing, the concurrency hint, the slab sums of the
// weight-gradient GEMMs.  Tile sizes are the same in both arithmetic modes (128 x 128 outputs per
// workgroup), so the slab counts and workspace sizes are too.
#pragma once
#include "pcb_common.h"

enum { PCB_PRO_PLAIN = 0, PCB_PRO_BNACT = 1, PCB_PRO_DY = 2, PCB_PRO_DY_POOL = 3 };

constexpr int PCB_NT_BM = 128, PCB_NT_BN = 128;  // output tile of a gemm_nt workgroup
constexpr int PCB_TN_BM = 128, PCB_TN_BN = 128, PCB_TN_RS = 32;  // gemm_tn: output tile, rows per stage
constexpr int PCB_MAX_SLABS = 768;  // upper bound of a gemm_nt launch's workgroups along x (= statistics slabs)

// CUs another kernel is known to occupy right now (pcb_set_concurrency_hint); read ONCE per entry
// point and passed down, so that one call sizes its grid and its slab count from the same value.
int pcb_busy_cus();

// Preferred number of workgroups along x (= statistics slabs) of a gemm_nt launch: persistent over
// row tiles, as many as are resident at once -- 2 per CU -- less what `busy_cus` other CUs are taken.  Never more than
// PCB_MAX_SLABS, never more than the row tiles.
long pcb_nt_grid_x(int pro, long R, int N, int busy_cus);

// The grid a launch actually uses: the caller's slab capacity caps the preference
// (nparts <= 0: no slabs are written, the preference stands).
inline long pcb_nt_grid_for(int pro, long R, int N, int busy_cus, int nparts)
{
    const long want = pcb_nt_grid_x(pro, R, N, busy_cus);
    return (nparts > 0 && nparts < want) ? nparts : want;
}

// Row splits of the weight-gradient GEMM: about `target` workgroups, at least 8 stages of work
// each.  Returns the split count; *rows_per_split is a multiple of PCB_TN_RS.
long pcb_tn_splits(long R, int M, int N, long *rows_per_split, long target);

// dW[M, k] (real weight layout, see real_column) = sum of `splits` slabs part[s][M*N], in slab
// order.  Runs right away, or -- between pcb_defer_reduces_begin and _flush on this thread -- is
// parked and run with the other parked sums in one launch.  quantum: 8 (bf16 rows) or 4 (fp32 rows).
int pcb_reduce_slabs(const float *part, int splits, long elems, float *dW, int N, int out_cols, int out_perm,
                     int quantum, hipStream_t st);

// pcb_reduce_slabs with slabs `stride` floats apart and, optionally, `vlen` further sums per slab stored behind
// its [M,N] part (they go to `vec`): the bias gradient of pcb_gemm_tn_bias_bf16.
int pcb_reduce_slabs_vec(const float *part, int splits, long stride, long elems, float *dW, int N, int out_cols,
                         int out_perm, int quantum, float *vec, int vlen, hipStream_t st);
