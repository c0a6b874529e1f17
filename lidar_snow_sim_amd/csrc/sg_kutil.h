// sg_kutil.h -- small device helpers shared by the kernel files (snowgpu_kernels.hip, snowgpu_rows.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "sg_common.h"
#include "sg_beam.h"

__device__ __forceinline__ int sg_find_frame(const int64_t *__restrict__ off, int n_frames, int64_t g)
{
    int lo = 0, hi = n_frames - 1;   // largest f with off[f] <= g
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (off[mid] <= g) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__device__ __forceinline__ unsigned long long sg_lanemask_lt()
{
    return (1ull << (threadIdx.x & 63)) - 1ull;
}

// ------------------------------------------------------------------------------------------------
// Frame of a sorted position.
__device__ __forceinline__ int sg_frame_of(const SgBeamArgs &a, int64_t g)
{
    if (a.uniform_rows > 0) {                    // equal-sized frames: no search (float estimate, integer fix-up)
        const unsigned rows_u = (unsigned)a.uniform_rows, gu = (unsigned)g;
        int fe = (int)((float)gu * a.inv_uniform_rows);
        if (fe >= a.n_frames) fe = a.n_frames - 1;
        while (fe > 0 && gu < (unsigned)fe * rows_u) --fe;
        while (fe + 1 < a.n_frames && gu >= (unsigned)(fe + 1) * rows_u) ++fe;
        return fe;
    }
    return sg_find_frame(a.frame_off, a.n_frames, g);
}

// Row of sorted position g of frame f: the input row itself when the frame came channel-sorted, else the sorted copy.
template <typename T>
__device__ __forceinline__ const T *sg_row(const SgBeamArgs &a, int f, int64_t g)
{
    return (const T *)(a.frame_unsorted[f] ? a.srows : a.rows) + g * 5;
}

// intensity_diff_sum (simulation.py:170, :512): one atomic per wave and frame, not one per beam (same-address atomics
// from every lane serialise in L2).  Every lane of the wave must call this.
__device__ __forceinline__ void sg_add_diff2(unsigned long long *diff2, bool live, int f, long long d2)
{
    unsigned long long todo = __ballot(live && d2 != 0);
    if (!todo) return;
    // addends below 2^24 in size (they are: twice an intensity difference): the wave's sum fits an int and a DPP prefix sum folds it;
    // anything else takes the 64-bit butterfly
    const bool small = __ballot(live && (d2 >= (1 << 24) || d2 <= -(1 << 24))) == 0;
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int fl = __builtin_amdgcn_readlane(f, leader);
        const bool mine = live && f == fl;
        long long part;
        if (small) part = (long long)__builtin_amdgcn_readlane(sg_wave_incl_add(mine ? (int)d2 : 0), 63);
        else {
            part = mine ? d2 : 0;
            for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
        }
        if ((int)(threadIdx.x & 63) == leader && part != 0) atomicAdd(&diff2[fl], (unsigned long long)part);
        todo &= ~__ballot(mine);
    }
}

