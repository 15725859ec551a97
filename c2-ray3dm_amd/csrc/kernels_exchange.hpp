// Device code, part 4 of 4: packing the sources' sub-boxes for the sparse exchange of the rates.  Included by exchange.hip only.
#pragma once
#include "kernels_common.hpp"

namespace c2r {

// ---- sparse exchange of the rates (cold regime, big meshes) --------------------------------------------------------
// evolve.F90:599 all-reduces the whole N^3 phih_grid after every pass, also while the rates are non-zero only inside a few
// sub-boxes.  Every rank knows every source's final sub-box (one small all-reduce of the sub-box counts), so all ranks agree
// on the same list of boxes: pack the rates of those boxes (box after box, in source order), all-reduce the packed
// buffer, write it back.  A cell of overlapping boxes travels once per box; the collective may sum the copies in different
// orders (RCCL's ring / tree order depends on the element's offset), so they can come back differing in the last bit.  The
// write-back therefore takes the MAXIMUM of what the cell holds and every copy -- rates are non-negative, whose f64 bit
// patterns order like unsigned integers, and a sum over the ranks is never below this rank's own addend (rounding is
// monotone) -- one 64-bit atomic max per copy: the same value on every rank whichever copy lands last, and the all-reduce's
// own sum wherever the copies agree.
struct BoxDesc { int c[3]; int nbox; long long off; };      // wrapped source cell, final sub-box count, first packed element
template <bool UNPACK>
__global__ __launch_bounds__(256) void k_pack_boxes(int n0, int n1, int n2, int hl0, int hl1, int hl2, int hr0, int hr1, int hr2,
                                                    int subbox, const BoxDesc *__restrict__ box, double *grid, double *packed)
{
    const BoxDesc b = box[blockIdx.y];
    if (b.nbox <= 0) return;
    const int ext = subbox * b.nbox;
    const int l0 = min(ext, hl0), l1 = min(ext, hl1), l2 = min(ext, hl2);
    const int e0 = l0 + min(ext, hr0) + 1, e1 = l1 + min(ext, hr1) + 1, e2 = l2 + min(ext, hr2) + 1;
    const long long vol = (long long)e0 * e1 * e2;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < vol; t += (long long)gridDim.x * 256) {
        const int i = (int)(t % e0), j = (int)((t / e0) % e1), k = (int)(t / ((long long)e0 * e1));
        const unsigned c0 = wrap_pos(b.c[0], n0, i - l0), c1 = wrap_pos(b.c[1], n1, j - l1), c2 = wrap_pos(b.c[2], n2, k - l2);
        const size_t id = (size_t)c0 + (size_t)n0 * ((size_t)c1 + (size_t)n1 * (size_t)c2);
        if (UNPACK) atomicMax(reinterpret_cast<unsigned long long *>(grid) + id, (unsigned long long)__double_as_longlong(packed[b.off + t]));
        else packed[b.off + t] = grid[id];
    }
}

}  // namespace c2r
