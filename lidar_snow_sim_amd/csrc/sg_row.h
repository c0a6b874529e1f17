// sg_row.h -- an output row of augment() from the ORIGINAL row and the beam's 4-byte result record (simulation.py:176-180, :516): what the
// compaction kernels (snowgpu_kernels.hip) and the CPU twin of the per-beam path (snowcpu.cpp) both do.  Device code that also compiles for
// the host (see sg_beam.h).
#pragma once
#include "sg_beam.h"

// ------------------------------------------------------------------------------------------------
// Output row of a sorted position, rebuilt from its ORIGINAL row and its result record (simulation.py:160-192, :516):
// unchanged rows keep their coordinates and get np.round(intensity); attenuated rows (label 1) the new intensity; scattered
// rows (label 2) move to d_max on their ray; rows of channels without a laser keep their channel value in column 4 (Q5).
// dd = the ORIGINAL range in the row dtype (simulation.py:465).
template <typename T> struct SgRow { T x, y, z, i, lab, dd; };

template <typename T>
__device__ __forceinline__ SgRow<T> sg_rebuild_row(const T *__restrict__ row, uint32_t rec)
{
    SgRow<T> r;
    const T px = row[0], py = row[1], pz = row[2], pint = row[3], pch = row[4];
    if constexpr (sizeof(T) == 4) r.dd = sqrtf((px * px + py * py) + pz * pz);
    else r.dd = sqrt((px * px + py * py) + pz * pz);
    r.x = px; r.y = py; r.z = pz;
    const int label = (int)((rec >> SG_REC_LABEL_SHIFT) & 3u);
    if (label == 0) {
        if constexpr (sizeof(T) == 4) r.i = rintf(pint); else r.i = rint(pint);      // :516 np.round (half to even)
        r.lab = (rec & SG_REC_COPY) ? pch : (T)0;
    } else {
        r.i = (T)(int)(rec & 255u);
        r.lab = (T)label;
        if (label == 2) {
            const double scale = sg_scatter_scale((int)((rec >> SG_REC_K_SHIFT) & 2047u), (double)r.dd);   // :176
            r.x = (T)((double)px * scale); r.y = (T)((double)py * scale); r.z = (T)((double)pz * scale);   // :178-180
        }
    }
    return r;
}

