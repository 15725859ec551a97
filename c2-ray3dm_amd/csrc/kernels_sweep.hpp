// Device code, part 2 of 4: the short-characteristics sweep -- source cells, one Chebyshev shell of every active source
// (exact and tolerance mode), look-ahead pairs, the fused first sub-boxes, the per-cell entry, the loss sums, the
// sub-box decision, and the per-pass helpers (n_HI preparation, the (x,y) transposes, the ordered Gamma sum).
// Included by sweep.hip ONLY (it defines non-template kernels).
#pragma once
#include "kernels_common.hpp"

namespace c2r {

// ---- source cells (q = 0) ------------------------------------------------------------------
// evolve_point.F90:151-160 (source cell) + the common tail of evolve0D, for source s; ltab: log10_tab's table (LDS or global).
// on_surface: degenerate meshes only, the source cell itself sits on the sub-box surface
template <int EXT>
__device__ __forceinline__ void source_cell(const KParams &p, const v2f64 *__restrict__ ltab, const int s, const bool on_surface,
                                            double *loss_acc, double *dbg_cdout)
{
    const int i = pmod(p.srcpos[3 * s + 0] - 1, p.n[0]);
    const int j = pmod(p.srcpos[3 * s + 1] - 1, p.n[1]);
    const int k = pmod(p.srcpos[3 * s + 2] - 1, p.n[2]);
    const size_t id = (size_t)i + (size_t)p.n[0] * ((size_t)j + (size_t)p.n[1] * (size_t)k);
    const double nhi = p.nhi[id];
    const double cd_in = 0.0;
    const double path = 0.5 * step_of(p).dr[0];
    const double vol_ph = step_of(p).dr[0] * step_of(p).dr[1] * step_of(p).dr[2];
    const double cd_out = cd_in + nhi * path;
    // plane q=0 of every face is the single cell (0,0)
    for (int f = 0; f < 6; ++f)
        p.planes[((size_t)s * 2 + 0) * 6 * p.PP + (size_t)f * p.PP + (size_t)p.R * p.P + p.R] = cd_out;
    if (dbg_cdout) dbg_cdout[id] = cd_out;
    const double nflux = p.normflux[s];
    double p_out = 0.0, gamma = 0.0;
    if (nflux > 0.0) {      // cd_in = 0 is never above max_coldensh
        double heat = 0.0;
        gamma = photoion<EXT>(p, ltab, cd_in, cd_out, vol_ph, nflux, p_out, &heat, (EXT & 2) ? p.normflux_x[s] : 0.0) / nhi;
        if (!p.gbox) atomicAdd(&p.phih[id], gamma);
        if ((EXT & 1) && !p.gbox && heat != 0.0) atomicAdd(&p.heat[id], heat);       // evolve_point.F90:285-286
        if ((EXT & 1) && p.gbox) p.gbox_h[(size_t)s * 2 * ((size_t)p.n[0] * p.n[1] * p.n[2]) + id] = heat;
    } else if ((EXT & 1) && p.gbox) p.gbox_h[(size_t)s * 2 * ((size_t)p.n[0] * p.n[1] * p.n[2]) + id] = 0.0;
    if (p.gbox) p.gbox[(size_t)s * 2 * ((size_t)p.n[0] * p.n[1] * p.n[2]) + id] = gamma;
    if (on_surface) loss_acc[s] += p_out * step_of(p).vol / vol_ph;
}

// One thread per source (the fused first sub-box does the same itself: BoxArgs::source_cell).
template <int EXT>
__global__ void k_source_cells(KParams p, int nsrc, const int *active, int boxR0, int boxR1, int boxR2,
                               int boxL0, int boxL1, int boxL2, double *loss_acc, double *dbg_cdout)
{
    __shared__ v2f64 s_log[kLogTab];                          // blocks of one wave
    const v2f64 *ltab = wave_log_table(p.logtab, s_log);
    const int sl = blockIdx.x * blockDim.x + threadIdx.x;
    if (sl >= nsrc) return;
    source_cell<EXT>(p, ltab, active[sl], boxR0 == 0 || boxR1 == 0 || boxR2 == 0 || boxL0 == 0 || boxL1 == 0 || boxL2 == 0,
                      loss_acc, dbg_cdout);
}

// ---- one Chebyshev shell of every active source ------------------------------------------------
// evolve0D (evolve_point.F90:83-299) + cinterp (column_density.f90:29-271) + photoion_rates.
// faces: 0:+z 1:-z 2:+y 3:-y 4:+x 5:-x.  Plane coordinates (a,b): z-face (x,y); y-face (x,z);
// x-face (y,z).  A cell on an edge/corner of the cube belongs to the face of highest priority
// (z over y over x: the branch order of cinterp); its owner also stores it into the other
// faces' planes, which read it in shell q+1.
// grid = (tiles, 6 faces, active sources); a wave runs along plane axis a, which is the
// unit-stride axis of the arrays the face reads (x for z/y faces, y in the transposed replicas
// for x faces).
// DET: deterministic_rates mode (per-source Gamma grids instead of atomics); LLS: type_of_LLS (1,2,3)
// ---- one cell, in two phases so that a thread can interleave two cells ---------------------------------
// CellState: everything evolve0D computes for cell (a,b) of face `face` in shell sa.q of source s before it
// touches memory for writing; cell_commit: the stores, the photo-ionization rate, the atomic and the
// photon loss.  The four upstream corners arrive as values (c*) with their weight reciprocals
// r* = 1/max(0.6, c*sigma) (weightf, column_density.f90:276-293), because neighbouring cells share them.
struct CellState {                 // kept small: it is live while the thread's other cell is worked on
    unsigned id, o8;
    double nhi, cd_in, cd_out, vol_ph;
    bool stop_far;
};

// mesh-axis deltas of plane cell (a,b) on a face normal to `axis` at signed distance pd (block-uniform selects)
struct Delta3 { int d0, d1, d2; };
__device__ __forceinline__ Delta3 mesh_delta(int axis, int pd, int a, int b)
{
    Delta3 d;
    d.d0 = (axis == 0) ? pd : a;
    d.d1 = (axis == 2) ? b : ((axis == 1) ? pd : a);
    d.d2 = (axis == 2) ? pd : b;
    return d;
}
// plane offset of (a,b) in bytes and the in-range test of a plane coordinate against shell q-1
__device__ __forceinline__ unsigned plane_off8(const KParams &p, int a, int b)
{
    return ((unsigned)((int)__umul24((unsigned)(b + p.R), (unsigned)p.P) + (a + p.R))) * 8u;   // factors in [0, 2^24)
}

// XONLY: read n_HI and the LLS grid from the x-fastest arrays whatever the face (same values; the look-ahead recompute
// calls this with a per-lane face and must not pick a buffer per lane)
template <int LLS, int STREAM /* bits: see shell_rows_fast_core */, bool XONLY = false>
__device__ __forceinline__ CellState cell_state(const KParams &p, const ShellArgs &sa, const int face, const int s,
                                                const int a, const int b, const double c1v, const double c2v,
                                                const double c3v, const double c4v, const double r1, const double r2,
                                                const double r3, const double r4)
{
    CellState cs;
    const int q = sa.q;
    const int axis = 2 - (face >> 1);            // 2:z 1:y 0:x
    const int pd = (face & 1) ? -q : q;
    // mesh-axis deltas and the source coordinates seen along (a,b)
    const Delta3 dl = mesh_delta(axis, pd, a, b);
    const int s0 = p.srcpos[3 * s + 0], s1 = p.srcpos[3 * s + 1], s2 = p.srcpos[3 * s + 2];
    const int su = (axis == 0) ? s1 : s0;
    const int sv = (axis == 2) ? s1 : s2;
    const unsigned c0 = wrap_pos(p.srcw[3 * s + 0], p.n[0], dl.d0);
    const unsigned c1 = wrap_pos(p.srcw[3 * s + 1], p.n[1], dl.d1);
    const unsigned c2 = wrap_pos(p.srcw[3 * s + 2], p.n[2], dl.d2);
    // cell index in the array this face reads: x-fastest, or y-fastest in the transposed replicas
    // (block-uniform choice; 24-bit multiplies: every factor is below 2^24)
    const bool xf = !XONLY && (axis == 0);
    const unsigned ca = xf ? c1 : c0, cb = xf ? c0 : c1;
    const unsigned na = xf ? (unsigned)p.n[1] : (unsigned)p.n[0], nb = xf ? (unsigned)p.n[0] : (unsigned)p.n[1];
    cs.id = ca + __umul24(na, cb + __umul24(nb, c2));
    const unsigned ncell = (unsigned)p.n[0] * (unsigned)p.n[1] * (unsigned)p.n[2];
    const __amdgpu_buffer_rsrc_t r_x = make_rsrc(xf ? p.nhi_T : p.nhi, ncell * 8u);
    cs.nhi = buf_load_f64<STREAM == 1 ? C2R_NHI_AUX : 0>(r_x, cs.id * 8u);
    cs.o8 = plane_off8(p, a, b);

    // cinterp, generic in (a,b,pd): the three branches differ only by which axes play (u,v).
    // real(int) conversions of the reference are f32 but exact (|.| < 2^24): cvt i32->f64.
    const int sga = a < 0 ? -1 : 1, sgb = b < 0 ? -1 : 1;
    const int am = a - sga, bm = b - sgb;
    const double du = (double)a, dv = (double)b;
    const double uc = sa.alam * du + (double)su;
    const double vc = sa.alam * dv + (double)sv;
    // real(im)+0.5*sgni (column_density.f90:117) is a half-integer: one conversion of 2*im+sgni
    const double ddu = 2.0 * fabs(uc - 0.5 * (double)(2 * (su + am) + sga));
    const double ddv = 2.0 * fabs(vc - 0.5 * (double)(2 * (sv + bm) + sgb));
    const double w1 = ((1. - ddu) * (1. - ddv)) * r1;
    const double w2 = ((1. - ddv) * ddu) * r2;
    const double w3 = ((1. - ddu) * ddv) * r3;
    const double w4 = (ddu * ddv) * r4;
    double cdi = fdiv(c1v * w1 + c2v * w2 + c3v * w3 + c4v * w4, w1 + w2 + w3 + w4);
    if (q == 1 && (abs(a) == 1 || abs(b) == 1))
        cdi = (abs(a) == 1 && abs(b) == 1) ? p.sqrt3 * cdi : p.sqrt2 * cdi;
    double path = fsqrt(udiv(du * du + dv * dv, sa.dp2, sa.inv_dp2, p.exact_udiv) + 1.0);

    // evolve0D
    path = path * step_of(p).dr[0];
    // dist2 = xs*xs + ys*ys + zs*zs (evolve_point.F90:171-174); the term of the face's own axis
    // is the same for the whole launch (ShellStep::d2axis = (dr_axis * q)^2)
    const double xs = step_of(p).dr[0] * (double)dl.d0;
    const double ys = step_of(p).dr[1] * (double)dl.d1;
    const double zs = step_of(p).dr[2] * (double)dl.d2;
    const C2R_AS4 ShellStep &ss = shell_step(p, q);
    const double xx = (axis == 0) ? ss.d2axis[0] : xs * xs;
    const double yy = (axis == 1) ? ss.d2axis[1] : ys * ys;
    const double zz = (axis == 2) ? ss.d2axis[2] : zs * zs;
    const double dist2 = xx + yy + zz;
    cs.vol_ph = p.fourpi * dist2 * path;
    // LLS opacity (evolve_point.F90:186-196): homogeneous column, per-cell column (LLS_point), or a
    // hard barrier at R_max that only stops the transfer
    cs.cd_in = cdi;
    cs.stop_far = false;
    if (LLS == 3) {
        cs.stop_far = dist2 > p.R_max2;
    } else {
        const double lls = (LLS == 2) ? (double)(xf ? p.lls_T : p.lls)[cs.id] : step_of(p).coldensh_LLS;
        cs.cd_in = cdi + udiv(lls * path, step_of(p).dr[0], step_of(p).inv_dr0, step_of(p).exact_udiv_dr0);
    }
    cs.cd_out = cs.cd_in + cs.nhi * path;
    return cs;
}

// Returns the cell's photon-loss contribution (0 unless it lies on the sub-box surface).
template <bool DET, int LLS, bool STREAM, int EXT, bool STORE = true>
__device__ __forceinline__ double cell_commit(const KParams &p, const ShellArgs &sa, const v2f64 *__restrict__ ltab,
                                              const int face, const int s, const int a, const int b, const CellState &cs)
{
    double loss = 0.0;
    const int q = sa.q;
    const int axis = 2 - (face >> 1);
    const int pd = (face & 1) ? -q : q;
    const bool xf = (axis == 0);
    const unsigned ncell = (unsigned)p.n[0] * (unsigned)p.n[1] * (unsigned)p.n[2];
    const unsigned plane_bytes = (unsigned)p.PP * 8u;
    const double cd_out = cs.cd_out;
    // store into this face's plane and into the planes of the faces sharing the cell
    const __amdgpu_buffer_rsrc_t r_cur = make_rsrc(p.planes + ((size_t)s * 2 + sa.buf_cur) * 6 * p.PP, 6u * plane_bytes);
    constexpr int SA = STREAM ? C2R_STORE_AUX : 0;
    if (STORE) {                 // (off for the first shell of a look-ahead pair: nothing reads its planes)
    buf_store_f64<SA>(r_cur, (unsigned)face * plane_bytes + cs.o8, cd_out);
    if (axis == 2) {
        if (abs(a) == q)   // x-face (u=y=b, v=z=pd)
            buf_store_f64<SA>(r_cur, (a > 0 ? 4u : 5u) * plane_bytes + (unsigned)((pd + p.R) * p.P + (b + p.R)) * 8u, cd_out);
        if (abs(b) == q)   // y-face (u=x=a, v=z=pd)
            buf_store_f64<SA>(r_cur, (b > 0 ? 2u : 3u) * plane_bytes + (unsigned)((pd + p.R) * p.P + (a + p.R)) * 8u, cd_out);
    } else if (axis == 1) {
        if (abs(a) == q)   // x-face (u=y=pd, v=z=b)
            buf_store_f64<SA>(r_cur, (a > 0 ? 4u : 5u) * plane_bytes + (unsigned)((b + p.R) * p.P + (pd + p.R)) * 8u, cd_out);
    }
    }
    const Delta3 dl = mesh_delta(axis, pd, a, b);          // recomputed, not carried in CellState
    if (sa.dbg_cdout) {                                     // single-source test path: the N^3 coldensh_out
        const unsigned c0 = wrap_pos(p.srcw[3 * s + 0], p.n[0], dl.d0), c1 = wrap_pos(p.srcw[3 * s + 1], p.n[1], dl.d1),
                       c2 = wrap_pos(p.srcw[3 * s + 2], p.n[2], dl.d2);
        sa.dbg_cdout[c0 + (unsigned)p.n[0] * (c1 + (unsigned)p.n[1] * c2)] = cd_out;
    }

    const double nflux = p.normflux[s];
    double gamma = 0.0, heat = 0.0;
    if (!cs.stop_far && !(cs.cd_in > p.max_coldensh) && nflux > 0.0) {
        double p_out;
        gamma = rates_fast<EXT>(p, ltab, p.thick, cs.cd_in, cd_out, nflux, (EXT & 2) ? p.normflux_x[s] : 0.0, cs.vol_ph * cs.nhi, cs.vol_ph, p_out, heat);
        if (!DET && gamma != 0.0) atomicAdd(&(xf ? p.phih_T : p.phih)[cs.id], gamma);
        if ((EXT & 1) && !DET && heat != 0.0) atomicAdd(&(xf ? p.heat_T : p.heat)[cs.id], heat);      // evolve_point.F90:285-286
        if (sa.has_boundary) {
            const bool bnd = dl.d0 == sa.boxR[0] || dl.d1 == sa.boxR[1] || dl.d2 == sa.boxR[2] ||
                             dl.d0 == -sa.boxL[0] || dl.d1 == -sa.boxL[1] || dl.d2 == -sa.boxL[2];
            if (bnd) loss = fdiv(p_out * step_of(p).vol, cs.vol_ph);
        }
    }
    // deterministic mode: every visited cell records its rate (zero included) for k_gamma_reduce
    if (DET) p.gbox[((size_t)s * 2 + (xf ? 1 : 0)) * ncell + cs.id] = gamma;
    if (DET && (EXT & 1)) p.gbox_h[((size_t)s * 2 + (xf ? 1 : 0)) * ncell + cs.id] = heat;
    return loss;
}

__device__ __forceinline__ double weight_rcp(const KParams &p, double c) { return frcp(fmax(p.wfloor, c * p.sigma)); }

// One cell (a,b): four upstream corners of plane q-1 (zero weight and value outside |.| <= q-1: an
// out-of-range offset reads 0), state, commit.  Used by the fused first-sub-box kernel.
template <bool DET, int LLS, int GLC, int EXT>
__device__ __forceinline__ double shell_cell(const KParams &p, const ShellArgs &sa, const v2f64 *__restrict__ ltab,
                                             const int face, const int s, const int a, const int b)
{
    const int q = sa.q, qm = q - 1;
    const int sga = a < 0 ? -1 : 1, sgb = b < 0 ? -1 : 1;
    const int am = a - sga, bm = b - sgb;
    const unsigned plane_bytes = (unsigned)p.PP * 8u;
    const __amdgpu_buffer_rsrc_t r_prev =
        make_rsrc(p.planes + ((size_t)s * 2 + sa.buf_prev) * 6 * p.PP + (size_t)face * p.PP, plane_bytes);
    const bool ina = abs(a) <= qm, inam = abs(am) <= qm, inb = abs(b) <= qm, inbm = abs(bm) <= qm;
    const unsigned p8 = (unsigned)p.P * 8u;
    const unsigned o8 = plane_off8(p, a, b), da8 = (unsigned)(sga * 8), db8 = b < 0 ? 0u - p8 : p8;
    const double c1v = buf_load_f64<GLC>(r_prev, (inam && inbm) ? o8 - db8 - da8 : kOOB);
    const double c2v = buf_load_f64<GLC>(r_prev, (ina && inbm) ? o8 - db8 : kOOB);
    const double c3v = buf_load_f64<GLC>(r_prev, (inam && inb) ? o8 - da8 : kOOB);
    const double c4v = buf_load_f64<GLC>(r_prev, (ina && inb) ? o8 : kOOB);
    const CellState cs = cell_state<LLS, false>(p, sa, face, s, a, b, c1v, c2v, c3v, c4v, weight_rcp(p, c1v), weight_rcp(p, c2v),
                                         weight_rcp(p, c3v), weight_rcp(p, c4v));
    return cell_commit<DET, LLS, false, EXT>(p, sa, ltab, face, s, a, b, cs);
}

// kRows cells of one column: (a,b0), (a,b0+sgb), ... with sgb the sign class of all their rows (rows are
// grouped outward from 0 within each sign class, see FaceRect).  Cell k+1's upstream row is cell k's own
// row, so the group needs 2(kRows+1) corners instead of 4 kRows -- and as many of the seven-instruction
// weight reciprocals -- and everything that depends on `a` alone (its sign, the u-interpolation factor,
// the wrap of that mesh axis) is computed once; the cells' dependency chains interleave in one thread and
// a wave's start-up is paid once for 64 kRows cells.  Per-cell arithmetic is exactly shell_cell's.
// Same-box A/B at 256^3 x 1000: 1 row 231 ms, 2 rows 206, 3 rows 194, 4 rows 203 (86 VGPRs, occupancy 5).
#ifndef C2R_ROWS
#define C2R_ROWS 3
#endif
constexpr int kRows = C2R_ROWS;
#ifndef C2R_PAIR_LDS_TABLE
#define C2R_PAIR_LDS_TABLE 0        // 1: the pair kernels fill the per-wave LDS table like the single-shell kernels (experiments)
#endif
constexpr int kPairRows = 1;        // rows per thread of the second shell of a look-ahead pair (k_sweep_pair, k_sweep_pair_fast)
template <bool DET, int LLS, int STREAM, int EXT, bool STORE = true>
__device__ __forceinline__ double shell_rows(const KParams &p, const ShellArgs &sa, const v2f64 *__restrict__ ltab,
                                             const int face, const int s, const int a, const int b0, const int sgb,
                                             const int nvalid)
{
    const int q = sa.q, qm = q - 1;
    const int sga = a < 0 ? -1 : 1;
    const int am = a - sga;
    const unsigned plane_bytes = (unsigned)p.PP * 8u;
    const __amdgpu_buffer_rsrc_t r_prev =
        make_rsrc(p.planes + ((size_t)s * 2 + sa.buf_prev) * 6 * p.PP + (size_t)face * p.PP, plane_bytes);
    const bool ina = abs(a) <= qm, inam = abs(am) <= qm;
    const unsigned p8 = (unsigned)p.P * 8u;
    const unsigned da8 = (unsigned)(sga * 8), db8 = sgb < 0 ? 0u - p8 : p8;
    unsigned o = plane_off8(p, a, b0) - db8;                 // row b0 - sgb
    double vm[kRows + 1], va[kRows + 1], rm[kRows + 1], ra[kRows + 1];
#pragma unroll
    for (int r = 0; r <= kRows; ++r) {                       // rows b0-sgb, b0, ..., b0+(kRows-1)sgb
        const bool inr = abs(b0 + (r - 1) * sgb) <= qm;
        vm[r] = buf_load_f64<(STREAM & 1) ? C2R_PLANE_AUX : 0>(r_prev, (inam && inr) ? o - da8 : kOOB);
        va[r] = buf_load_f64<(STREAM & 1) ? C2R_PLANE_AUX : 0>(r_prev, (ina && inr) ? o : kOOB);
        o += db8;
    }
#pragma unroll
    for (int r = 0; r <= kRows; ++r) { rm[r] = weight_rcp(p, vm[r]); ra[r] = weight_rcp(p, va[r]); }
    const CellState c0 = cell_state<LLS, STREAM>(p, sa, face, s, a, b0, vm[0], va[0], vm[1], va[1], rm[0], ra[0], rm[1], ra[1]);
    const CellState c1 = cell_state<LLS, STREAM>(p, sa, face, s, a, b0 + sgb, vm[1], va[1], vm[2], va[2], rm[1], ra[1], rm[2], ra[2]);
#if C2R_ROWS >= 3
    const CellState c2 = cell_state<LLS, STREAM>(p, sa, face, s, a, b0 + 2 * sgb, vm[2], va[2], vm[3], va[3], rm[2], ra[2], rm[3], ra[3]);
#endif
#if C2R_ROWS >= 4
    const CellState c3 = cell_state<LLS, STREAM>(p, sa, face, s, a, b0 + 3 * sgb, vm[3], va[3], vm[4], va[4], rm[3], ra[3], rm[4], ra[4]);
#endif
    double loss = cell_commit<DET, LLS, (STREAM & 1) != 0, EXT, STORE>(p, sa, ltab, face, s, a, b0, c0);
    if (nvalid > 1) loss = loss + cell_commit<DET, LLS, (STREAM & 1) != 0, EXT, STORE>(p, sa, ltab, face, s, a, b0 + sgb, c1);
#if C2R_ROWS >= 3
    if (nvalid > 2) loss = loss + cell_commit<DET, LLS, (STREAM & 1) != 0, EXT, STORE>(p, sa, ltab, face, s, a, b0 + 2 * sgb, c2);
#endif
#if C2R_ROWS >= 4
    if (nvalid > 3) loss = loss + cell_commit<DET, LLS, (STREAM & 1) != 0, EXT, STORE>(p, sa, ltab, face, s, a, b0 + 3 * sgb, c3);
#endif
    return loss;
}

// Look-ahead of the exact mode (k_sweep_pair; see lookahead_cd_out of the fast mode below for the idea): the column density that
// shell sq.q leaves in plane `face` at (a, b), recomputed from the planes of shell sq.q - 1 with cell_state -- the arithmetic of
// the launch that owns the cell, in the owner's geometry.  Straight-line code: a thread's four recomputes share their waits.
template <int LLS, bool STREAM>
__device__ __forceinline__ double lookahead_cd_out_exact(const KParams &p, const ShellArgs &sq, const int face, const int s,
                                                         const int a, const int b)
{
    const int q = sq.q, qm = q - 1;
    const bool inside = abs(a) <= q && abs(b) <= q;           // else a zero-weight corner (an OOB load reads 0)
    const int axis = 2 - (face >> 1);
    const int pd = (face & 1) ? -q : q;
    int fo = face, ao = a, bo = b;                            // the owner of the cell and its coordinates there
    if (axis == 1) { if (abs(b) == q) { fo = b > 0 ? 0 : 1; ao = a; bo = pd; } }
    else if (axis == 0) {
        if (abs(b) == q) { fo = b > 0 ? 0 : 1; ao = pd; bo = a; }
        else if (abs(a) == q) { fo = a > 0 ? 2 : 3; ao = pd; bo = b; }
    }
    const int sga = ao < 0 ? -1 : 1, sgb = bo < 0 ? -1 : 1;
    const int am = ao - sga, bm = bo - sgb;
    const unsigned plane_bytes = (unsigned)p.PP * 8u;
    const __amdgpu_buffer_rsrc_t r_prev = make_rsrc(p.planes + ((size_t)s * 2 + sq.buf_prev) * 6 * p.PP, 6u * plane_bytes);
    const bool ina = abs(ao) <= qm, inam = abs(am) <= qm, inb = abs(bo) <= qm, inbm = abs(bm) <= qm;
    const unsigned base = (unsigned)fo * plane_bytes;
    constexpr int PA = STREAM ? C2R_PLANE_AUX : 0;
    const double c1v = buf_load_f64<PA>(r_prev, (inam && inbm) ? base + plane_off8(p, am, bm) : kOOB);
    const double c2v = buf_load_f64<PA>(r_prev, (ina && inbm) ? base + plane_off8(p, ao, bm) : kOOB);
    const double c3v = buf_load_f64<PA>(r_prev, (inam && inb) ? base + plane_off8(p, am, bo) : kOOB);
    const double c4v = buf_load_f64<PA>(r_prev, (ina && inb) ? base + plane_off8(p, ao, bo) : kOOB);
    const CellState cs = cell_state<LLS, STREAM, true>(p, sq, fo, s, ao, bo, c1v, c2v, c3v, c4v, weight_rcp(p, c1v),
                                                       weight_rcp(p, c2v), weight_rcp(p, c3v), weight_rcp(p, c4v));
    return inside ? cs.cd_out : 0.0;
}
// one cell of the second shell of a pair: its four upstream corners recomputed
template <bool DET, int LLS, bool STREAM, int EXT>
__device__ __forceinline__ double shell_cell_look(const KParams &p, const ShellArgs &sa, const ShellArgs &sq,
                                                  const v2f64 *__restrict__ ltab, const int face, const int s, const int a, const int b)
{
    const int sga = a < 0 ? -1 : 1, sgb = b < 0 ? -1 : 1;
    const int am = a - sga, bm = b - sgb;
    const double c1v = lookahead_cd_out_exact<LLS, STREAM>(p, sq, face, s, am, bm);
    const double c2v = lookahead_cd_out_exact<LLS, STREAM>(p, sq, face, s, a, bm);
    const double c3v = lookahead_cd_out_exact<LLS, STREAM>(p, sq, face, s, am, b);
    const double c4v = lookahead_cd_out_exact<LLS, STREAM>(p, sq, face, s, a, b);
    const CellState cs = cell_state<LLS, STREAM>(p, sa, face, s, a, b, c1v, c2v, c3v, c4v, weight_rcp(p, c1v), weight_rcp(p, c2v),
                                                 weight_rcp(p, c3v), weight_rcp(p, c4v));
    return cell_commit<DET, LLS, STREAM, EXT>(p, sa, ltab, face, s, a, b, cs);
}

// STREAM: the non-temporal cache policy above (the host turns it on for meshes whose n_HI array outgrows the L2s)
// LOOK (sq = the previous shell's arguments): one row per thread (FaceRect built for kPairRows), corners recomputed
// (src < 0: the source is sa.active[sl]; else src, and sl only indexes the loss partials of a shell on the sub-box surface)
template <bool DET, int LLS, int STREAM, int EXT, int LOOK = 0, bool STORE = true>
__device__ __forceinline__ void sweep_tile(const KParams &p, const ShellArgs &sa, const FaceRect &fr, const v2f64 *ltab,
                                           double *sm, const int face, const int tile, const int sl, const ShellArgs &sq,
                                           const int src = -1)
{
    double loss = 0.0;
    const unsigned t = (unsigned)tile * kBlock + threadIdx.x;
    const unsigned bi = fr.magic ? __umulhi(t, fr.magic) : t;       // row-group index
    if (tile < fr.ntiles && bi < (unsigned)fr.npr) {
        const int a = fr.a_lo + (int)(t - __umul24(bi, (unsigned)fr.wa));
        // rows are grouped outward from 0 within each sign class: (0..kRows-1), ... then (-1..-kRows), ...
        const bool pos = bi < (unsigned)fr.pp;
        constexpr int NR = LOOK ? kPairRows : kRows;
        const int k0 = NR * (pos ? (int)bi : (int)bi - fr.pp);
        const int sgb = pos ? 1 : -1;
        const int b0 = pos ? k0 : -1 - k0;
        const int left = pos ? (fr.b_lo + fr.wb - b0) : (b0 - fr.b_lo + 1);     // rows from b0 to the end of the class
        const int s = src < 0 ? sa.active[sl] : src;
        if (LOOK) loss = shell_cell_look<DET, LLS, (STREAM & 1) != 0, EXT>(p, sa, sq, ltab, face, s, a, b0);
        else loss = shell_rows<DET, LLS, STREAM, EXT, STORE>(p, sa, ltab, face, s, a, b0, sgb, min(left, kRows));
    }
    if (sa.has_boundary) {
        const double tot = block_sum_256(loss, sm);
        if (threadIdx.x == 0)
            sa.loss_partial[((size_t)sl * 6 + face) * sa.tiles_max + tile] = tot;
    }
}

template <bool DET, int LLS, bool STREAM, int EXT>
__global__ __launch_bounds__(kBlock) void k_sweep_shell(KParams p, ShellArgs sa)
{
    __shared__ double sm[16];
    __shared__ v2f64 s_log[kBlock];                      // kBlock/64 waves x 64 entries
    const int face = blockIdx.y;
    const int tile = blockIdx.x;
    const int nact = *sa.n_active;
    if ((int)blockIdx.z >= nact) return;         // block-uniform: this source retired after the launch was sized
    const FaceRect fr = sa.face[face];
    if (tile >= fr.ntiles && !sa.has_boundary) return;   // block-uniform: nothing to do and no partial to write
    const v2f64 *ltab = wave_log_table(p.odtab, s_log);      // (table positions: rates_fast)
    sweep_tile<DET, LLS, STREAM, EXT>(p, sa, fr, ltab, sm, face, tile, (int)blockIdx.z, sa);
}

// The look-ahead pair of the exact mode (see k_sweep_pair_fast below): shell sa.q in blockIdx.y 0..5 (planes not stored),
// shell sb.q = sa.q + 1 in 6..11 (corners recomputed, one row per thread).
template <bool DET, int LLS, bool STREAM, int EXT>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(1, 4)))
void k_sweep_pair(KParams p, ShellArgs sa, ShellArgs sb)
{
    __shared__ double sm[16];
    __shared__ v2f64 s_log[kBlock];
    const bool second = blockIdx.y >= 6;
    const int face = second ? (int)blockIdx.y - 6 : (int)blockIdx.y;
    const int tile = blockIdx.x;
    const int nact = *sa.n_active;
    if ((int)blockIdx.z >= nact) return;
    const v2f64 *ltab = (C2R_PAIR_LDS_TABLE) ? wave_log_table(p.odtab, s_log) : p.odtab;     // see k_sweep_pair_fast
    if (second) {
        const FaceRect fr = sb.face[face];
        if (tile >= fr.ntiles) return;
        sweep_tile<DET, LLS, STREAM, EXT, 1, true>(p, sb, fr, ltab, sm, face, tile, (int)blockIdx.z, sa);
    } else {
        const FaceRect fr = sa.face[face];
        if (tile >= fr.ntiles) return;
        sweep_tile<DET, LLS, STREAM, EXT, 0, false>(p, sa, fr, ltab, sm, face, tile, (int)blockIdx.z, sa);
    }
}

// ==== tolerance ("fast") mode of the sweep =======================================================
// Same physics, same schedule, same integer decisions; the f64 arithmetic of one (cell, source) is
// re-associated where algebra allows, giving up bit-identity of the column densities with the
// Fortran (c2r_params.sweep_mode = 1; the exact kernel above stays the default).  Against the exact
// kernel: ~1e-13 relative on column densities and rates (stated and tested in tests/: integers exact,
// xh 1e-9).  What changes, with the reference lines each form restates:
//  * cinterp weights (column_density.f90:112-140): dx = 2|xc-(im+sgn/2)| is 1-|a|/q identically, so the
//    four weights factor into column and row parts; with t = c/max(0.6,c sigma), r = 1/max(0.6,c sigma)
//    per corner, cdensi = [omv(omu t1+ddu t2)+ddv(omu t3+ddu t4)] / [same in r]: the column sums of a row
//    are shared by the two cells above and below it (3 rows per thread: 4 rows of sums for 3 cells);
//  * reciprocals by v_rcp_f64 + one Newton step (2^-48) instead of the correctly rounded quotient;
//  * path = sqrt(q^2+a^2+b^2) dr/q from the exact integer (column_density.f90:168), dist2 by two FMAs
//    (evolve_point.F90:171-174), Gamma = (T_in-T_out) NormFlux / (4 pi dist2 n_HI path) with one
//    reciprocal (radiation_photoionrates.F90:262-263, evolve_point.F90:262);
//  * the table position 1+(log10 tau-minlogtau)/dlogtau (radiation_photoionrates.F90:195-199) comes
//    straight out of the log evaluation: the LDS table holds positions instead of logarithms.
__device__ __forceinline__ double sqrt_pos(double x)    // sqrt(x), x >= 1 normal: Goldschmidt step + one correction (~1 ulp)
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    const double h = 0.5 * y;
    g = __builtin_fma(g, __builtin_fma(-h, g, 0.5), g);
    return __builtin_fma(__builtin_fma(-g, g, x), h, g);
}
// Row sums over the two upstream columns am, a of one upstream row: R = omu r(am) + ddu r(a), T likewise with t = c r
// (r = 1/max(0.6, c sigma), weightf of column_density.f90:276-293)
__device__ __forceinline__ void row_sum_fast(const KParams &p, const double omu, const double ddu, const double vm, const double va,
                                             double &R, double &T)
{
    const double rm = rcp1(fmax(p.wfloor, vm * p.sigma)), ra = rcp1(fmax(p.wfloor, va * p.sigma));
    R = __builtin_fma(omu, rm, ddu * ra);
    T = __builtin_fma(omu, vm * rm, ddu * (va * ra));
}
// cinterp + path + the cell's own column for cell (a, b) of a face in shell sa.q, from the row sums of its upstream rows
// b - sgb (lo) and b (hi).  ONE piece of arithmetic for the shell kernel and for the look-ahead recompute of the same cell
// (lookahead_cd_out): both must produce the same bits.
struct CellCd { double cd_in, cd_out, pq, path, np; };
template <int LLS>
__device__ __forceinline__ CellCd cell_cd_fast(const KParams &p, const ShellArgs &sa, const int a, const int a2, const int b,
                                               const double Rlo, const double Tlo, const double Rhi, const double Thi,
                                               const double nhi, const double lls_cell /* LLS == 2: the cell's LLS_grid value */)
{
    CellCd c;
    const int q = sa.q;
    const double omv = (double)abs(b) * sa.inv_q, ddv = 1.0 - omv;      // weights of rows b-sgb and b
    const double den = __builtin_fma(omv, Rlo, ddv * Rhi);
    const double num = __builtin_fma(omv, Tlo, ddv * Thi);
    double cdi = num * rcp1(den);
    if (q == 1 && (abs(a) == 1 || abs(b) == 1))
        cdi = (abs(a) == 1 && abs(b) == 1) ? p.sqrt3 * cdi : p.sqrt2 * cdi;
    c.pq = sqrt_pos((double)(q * q + a2 + b * b));                      // |delta| in cells
    const C2R_AS4 ShellStep &ss = shell_step(p, q);
    c.path = c.pq * ss.path_scale;
    if (LLS == 3) c.cd_in = cdi;
    else if (LLS == 2) c.cd_in = __builtin_fma(lls_cell * sa.inv_q, c.pq, cdi);
    else c.cd_in = __builtin_fma(ss.lls_scale, c.pq, cdi);
    c.np = nhi * c.path;                                                // n_HI path: the cell's own column
    c.cd_out = c.cd_in + c.np;
    return c;
}

// Look-ahead (few sources, k_sweep_pair_fast): the column density that shell sq.q leaves in plane `face` at plane
// coordinates (a, b), recomputed from the planes of shell sq.q - 1 instead of read -- so that shell q+1 can run in the
// SAME launch as shell q (both depend on shell q-1 only): with few sources the GPU is idle and a launch is pure
// latency, the redundant arithmetic is free and the chain of dependent launches is halved.  A plane also holds the
// cells that a neighbouring face owns (the edge rows / columns the owner stores across, shell_rows_fast_core): those
// are recomputed in the owner's geometry.  Same arithmetic as the owner's own launch (row_sum_fast, cell_cd_fast).
template <int LLS, bool STREAM>
__device__ __forceinline__ double lookahead_cd_out(const KParams &p, const ShellArgs &sq, const int face, const int s,
                                                   const int a, const int b)
{
    const int q = sq.q, qm = q - 1;
    // straight-line code (no early exit): a thread recomputes 8 such cells and all their loads should be in flight together
    const bool inside = abs(a) <= q && abs(b) <= q;           // else: outside shell q's plane, a zero-weight corner (an OOB load reads 0)
    const int axis = 2 - (face >> 1);
    const int pd = (face & 1) ? -q : q;
    // the owner of the cell and the cell's coordinates on the owner's face
    int fo = face, ao = a, bo = b;
    if (axis == 1) { if (abs(b) == q) { fo = b > 0 ? 0 : 1; ao = a; bo = pd; } }
    else if (axis == 0) {
        if (abs(b) == q) { fo = b > 0 ? 0 : 1; ao = pd; bo = a; }
        else if (abs(a) == q) { fo = a > 0 ? 2 : 3; ao = pd; bo = b; }
    }
    const int sga = ao < 0 ? -1 : 1, sgb = bo < 0 ? -1 : 1;
    const int am = ao - sga, bm = bo - sgb;
    const unsigned plane_bytes = (unsigned)p.PP * 8u;
    const __amdgpu_buffer_rsrc_t r_prev = make_rsrc(p.planes + ((size_t)s * 2 + sq.buf_prev) * 6 * p.PP, 6u * plane_bytes);
    const bool ina = abs(ao) <= qm, inam = abs(am) <= qm, inb = abs(bo) <= qm, inbm = abs(bm) <= qm;
    const unsigned base = (unsigned)fo * plane_bytes;
    constexpr int PA = STREAM ? C2R_PLANE_AUX : 0;
    const double v_mm = buf_load_f64<PA>(r_prev, (inam && inbm) ? base + plane_off8(p, am, bm) : kOOB);
    const double v_am = buf_load_f64<PA>(r_prev, (ina && inbm) ? base + plane_off8(p, ao, bm) : kOOB);
    const double v_mb = buf_load_f64<PA>(r_prev, (inam && inb) ? base + plane_off8(p, am, bo) : kOOB);
    const double v_ab = buf_load_f64<PA>(r_prev, (ina && inb) ? base + plane_off8(p, ao, bo) : kOOB);
    // the cell in the mesh: from its position on the plane it was asked for
    const Delta3 dl = mesh_delta(axis, pd, a, b);
    const unsigned c0 = wrap_pos(p.srcw[3 * s + 0], p.n[0], dl.d0), c1 = wrap_pos(p.srcw[3 * s + 1], p.n[1], dl.d1),
                   c2 = wrap_pos(p.srcw[3 * s + 2], p.n[2], dl.d2);
    const unsigned id = c0 + (unsigned)p.n[0] * (c1 + (unsigned)p.n[1] * c2);
    const unsigned ncell = (unsigned)p.n[0] * (unsigned)p.n[1] * (unsigned)p.n[2];
    const double nhi = buf_load_f64<STREAM ? C2R_NHI_AUX : 0>(make_rsrc(p.nhi, ncell * 8u), id * 8u);
    const double omu = (double)abs(ao) * sq.inv_q, ddu = 1.0 - omu;
    double Rlo, Tlo, Rhi, Thi;
    row_sum_fast(p, omu, ddu, v_mm, v_am, Rlo, Tlo);
    row_sum_fast(p, omu, ddu, v_mb, v_ab, Rhi, Thi);
    const double cd_out = cell_cd_fast<LLS>(p, sq, ao, ao * ao, bo, Rlo, Tlo, Rhi, Thi, nhi, LLS == 2 ? (double)p.lls[id] : 0.0).cd_out;
    return inside ? cd_out : 0.0;
}

// Everything a thread does for its NR rows once the upstream values are known: vm[r], va_[r] = the previous shell's
// column densities at columns am = a - sga and a of rows b0 - sgb, b0, ..., b0 + (NR-1) sgb.
// STORE: write the column densities into the planes (off for the first shell of a look-ahead pair: nothing reads them)
#ifndef C2R_XCD_KEEP_DBG
#define C2R_XCD_KEEP_DBG 0          // 1 (experiments): keep the debug path in the plane-ordered kernel (65 VGPRs, 7 waves)
#endif
// STREAM (here and in shell_rows_fast / sweep_tile_fast / sweep_tile): bit 0 -- non-temporal cache hints on the streams; bit 1 -- the
// plane-ordered mapping (k_sweep_shell_xcd): the n_HI loads keep the plain policy (they are to stay in the XCD's L2) and the
// coldensh_out debug path, which that mapping never runs with, is compiled out (63 VGPRs instead of 65: 8 waves per SIMD)
// STORE: 0 nothing, 1 the source's global planes, 2 the workgroup's LDS planes (k_sweep_box_fused: lds_cur, pitch kLdsP)
constexpr int kLdsR = kFusedQmaxK, kLdsP = 2 * kFusedQmaxK + 1, kLdsPP = kLdsP * kLdsP;
template <bool DET, int LLS, int STREAM, int NR, int EXT, int STORE = 1>
__device__ __forceinline__ double shell_rows_fast_core(const KParams &p, const ShellArgs &sa, const v2f64 *__restrict__ ltab,
                                                       const double *__restrict__ thick, const int face, const int s, const int a,
                                                       const int b0, const int sgb, const int nvalid,
                                                       const double (&vm)[NR + 1], const double (&va_)[NR + 1],
                                                       double *__restrict__ lds_cur = nullptr)
{
    const int q = sa.q;
    const int axis = 2 - (face >> 1);            // 2:z 1:y 0:x  (block-uniform)
    const int pd = (face & 1) ? -q : q;
    const bool xf = (axis == 0);
    const int ua = xf ? 1 : 0, va = (axis == 2) ? 1 : 2;      // mesh axes of the plane coordinates (a, b)
    const unsigned plane_bytes = (unsigned)p.PP * 8u;
    const unsigned p8 = (unsigned)p.P * 8u;
    const unsigned db8 = sgb < 0 ? 0u - p8 : p8;
    const unsigned o_first = plane_off8(p, a, b0);
    // cell indices: id = ca + base_p + stride_b * cb (x-fastest array, or the (x,y)-transposed replica for x faces)
    const unsigned na = (unsigned)p.n[ua], nmid = xf ? (unsigned)p.n[0] : (unsigned)p.n[1];
    const unsigned ca = wrap_pos(p.srcw[3 * s + ua], p.n[ua], a);
    const unsigned cp = wrap_pos(p.srcw[3 * s + axis], p.n[axis], pd);          // block-uniform
    const unsigned stride_b = (axis == 2) ? na : na * nmid;
    const unsigned base_p = (axis == 2) ? na * nmid * cp : na * cp;
    const unsigned ncell = (unsigned)p.n[0] * (unsigned)p.n[1] * (unsigned)p.n[2];
    const __amdgpu_buffer_rsrc_t r_x = make_rsrc(xf ? p.nhi_T : p.nhi, ncell * 8u);
    unsigned id[NR];
    double nhi[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        const unsigned cb = wrap_pos(p.srcw[3 * s + va], p.n[va], b0 + k * sgb);
        id[k] = ca + base_p + __umul24(stride_b, cb);
        nhi[k] = buf_load_f64<STREAM == 1 ? C2R_NHI_AUX : 0>(r_x, id[k] * 8u);
    }
    // column part of the interpolation and of the geometry
    const double omu = (double)abs(a) * sa.inv_q, ddu = 1.0 - omu;   // weights of columns am and a
    const int a2 = a * a;
    const double du2 = step_of(p).dr2[ua] * (double)a2;
    // row sums over the two columns
    double R[NR + 1], T[NR + 1];
#pragma unroll
    for (int r = 0; r <= NR; ++r) row_sum_fast(p, omu, ddu, vm[r], va_[r], R[r], T[r]);
    const double nflux = p.normflux[s];
    const double nflux_x = (EXT & 2) ? p.normflux_x[s] : 0.0;
    const __amdgpu_buffer_rsrc_t r_cur = make_rsrc(p.planes + ((size_t)s * 2 + sa.buf_cur) * 6 * p.PP, 6u * plane_bytes);
    constexpr int SA = (STREAM & 1) ? C2R_STORE_AUX : 0;
    const bool bnd_col = sa.has_boundary && (a == sa.boxR[ua] || a == -sa.boxL[ua] || pd == sa.boxR[axis] || pd == -sa.boxL[axis]);
    double loss = 0.0;
    unsigned o8 = o_first;
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        const int b = b0 + k * sgb;
        if (k < nvalid) {
            const CellCd cc = cell_cd_fast<LLS>(p, sa, a, a2, b, R[k], T[k], R[k + 1], T[k + 1], nhi[k],
                                                LLS == 2 ? (double)(xf ? p.lls_T : p.lls)[id[k]] : 0.0);
            const double path = cc.path, cd_in = cc.cd_in, np = cc.np, cd_out = cc.cd_out;
            const double dist2 = __builtin_fma(step_of(p).dr2[va], (double)(b * b), du2 + shell_step(p, q).d2axis[axis]);
            const bool stop = (LLS == 3) && dist2 > p.R_max2;
            // the cell's column density, also into the planes of the faces sharing the cell
            if (STORE == 1) {
            buf_store_f64<SA>(r_cur, (unsigned)face * plane_bytes + o8, cd_out);
            if (axis == 2) {
                if (abs(a) == q)
                    buf_store_f64<SA>(r_cur, (a > 0 ? 4u : 5u) * plane_bytes + (unsigned)((pd + p.R) * p.P + (b + p.R)) * 8u, cd_out);
                if (abs(b) == q)
                    buf_store_f64<SA>(r_cur, (b > 0 ? 2u : 3u) * plane_bytes + (unsigned)((pd + p.R) * p.P + (a + p.R)) * 8u, cd_out);
            } else if (axis == 1) {
                if (abs(a) == q)
                    buf_store_f64<SA>(r_cur, (a > 0 ? 4u : 5u) * plane_bytes + (unsigned)((b + p.R) * p.P + (pd + p.R)) * 8u, cd_out);
            }
            }
            if (STORE == 2) {            // the same entries of the workgroup's LDS planes (the next shell of this launch reads them)
                lds_cur[face * kLdsPP + (b + kLdsR) * kLdsP + (a + kLdsR)] = cd_out;
                if (axis == 2) {
                    if (abs(a) == q) lds_cur[(a > 0 ? 4 : 5) * kLdsPP + (pd + kLdsR) * kLdsP + (b + kLdsR)] = cd_out;
                    if (abs(b) == q) lds_cur[(b > 0 ? 2 : 3) * kLdsPP + (pd + kLdsR) * kLdsP + (a + kLdsR)] = cd_out;
                } else if (axis == 1) {
                    if (abs(a) == q) lds_cur[(a > 0 ? 4 : 5) * kLdsPP + (b + kLdsR) * kLdsP + (pd + kLdsR)] = cd_out;
                }
            }
            if ((C2R_XCD_KEEP_DBG || !(STREAM & 2)) && sa.dbg_cdout) {
                const Delta3 dl = mesh_delta(axis, pd, a, b);
                const unsigned c0 = wrap_pos(p.srcw[3 * s + 0], p.n[0], dl.d0), c1 = wrap_pos(p.srcw[3 * s + 1], p.n[1], dl.d1),
                               c2 = wrap_pos(p.srcw[3 * s + 2], p.n[2], dl.d2);
                sa.dbg_cdout[c0 + (unsigned)p.n[0] * (c1 + (unsigned)p.n[1] * c2)] = cd_out;
            }
            double gamma = 0.0, heat = 0.0;
            if (!stop && !(cd_in > p.max_coldensh) && nflux > 0.0) {
                const double area = p.fourpi * dist2;                             // vol_ph = area path
                double p_out;
                gamma = rates_fast<EXT>(p, ltab, thick, cd_in, cd_out, nflux, nflux_x, area * np, area * path, p_out, heat);
                if (!DET && gamma != 0.0) atomicAdd(&(xf ? p.phih_T : p.phih)[id[k]], gamma);
                if ((EXT & 1) && !DET && heat != 0.0) atomicAdd(&(xf ? p.heat_T : p.heat)[id[k]], heat);
                if (sa.has_boundary && (bnd_col || b == sa.boxR[va] || b == -sa.boxL[va]))
                    loss = loss + fdiv(p_out * step_of(p).vol, area * path);
            }
            if (DET) p.gbox[((size_t)s * 2 + (xf ? 1 : 0)) * ncell + id[k]] = gamma;
            if (DET && (EXT & 1)) p.gbox_h[((size_t)s * 2 + (xf ? 1 : 0)) * ncell + id[k]] = heat;
        }
        o8 += db8;
    }
    return loss;
}

// LOOK = 0: the upstream values are read from the previous shell's planes (sq unused: pass sa).  LOOK = 1 (sq = the previous
// shell's arguments, by reference -- a pointer to a kernel argument would force it into scratch): they are recomputed from
// the planes of the shell before it (lookahead_cd_out).
// LOOK = 2 (k_sweep_box_fused, shells after the launch's first): the four upstream column densities of every cell are read from
// the workgroup's LDS planes of the previous shell (lds_prev) -- the shell-to-shell hand-off stays on chip.
template <bool DET, int LLS, int STREAM, int NR, int EXT, int LOOK = 0, int STORE = 1>
__device__ __forceinline__ double shell_rows_fast(const KParams &p, const ShellArgs &sa, const v2f64 *__restrict__ ltab,
                                                  const double *__restrict__ thick,   // p.thick, or the block's LDS copy of it
                                                  const int face, const int s, const int a, const int b0, const int sgb,
                                                  const int nvalid, const ShellArgs &sq,
                                                  const double *__restrict__ lds_prev = nullptr, double *__restrict__ lds_cur = nullptr)
{
    const int q = sa.q, qm = q - 1;
    const int sga = a < 0 ? -1 : 1;
    const int am = a - sga;
    double vm[NR + 1], va_[NR + 1];
    if (LOOK == 2) {
        const bool ina = abs(a) <= qm, inam = abs(am) <= qm;
        const double *pl = lds_prev + face * kLdsPP + kLdsR * kLdsP + kLdsR;
#pragma unroll
        for (int r = 0; r <= NR; ++r) {                        // rows b0-sgb, b0, ..., b0+(NR-1)sgb
            const int row = b0 + (r - 1) * sgb;
            const bool inr = abs(row) <= qm;                   // (a corner outside the previous shell: weight 0, value 0, as an OOB load)
            vm[r] = (inam && inr) ? pl[row * kLdsP + am] : 0.0;
            va_[r] = (ina && inr) ? pl[row * kLdsP + a] : 0.0;
        }
    } else if (LOOK) {
#pragma unroll
        for (int r = 0; r <= NR; ++r) {                        // rows b0-sgb, b0, ..., b0+(NR-1)sgb
            // (rows beyond the thread's valid cells are computed too -- from periodic-wrapped, in-range addresses -- and unused)
            const int row = b0 + (r - 1) * sgb;
            vm[r] = lookahead_cd_out<LLS, (STREAM & 1) != 0>(p, sq, face, s, am, row);
            va_[r] = lookahead_cd_out<LLS, (STREAM & 1) != 0>(p, sq, face, s, a, row);
        }
    } else {
        const unsigned plane_bytes = (unsigned)p.PP * 8u;
        const __amdgpu_buffer_rsrc_t r_prev =
            make_rsrc(p.planes + ((size_t)s * 2 + sa.buf_prev) * 6 * p.PP + (size_t)face * p.PP, plane_bytes);
        const bool ina = abs(a) <= qm, inam = abs(am) <= qm;
        const unsigned p8 = (unsigned)p.P * 8u;
        const unsigned da8 = (unsigned)(sga * 8), db8 = sgb < 0 ? 0u - p8 : p8;
        unsigned o = plane_off8(p, a, b0) - db8;                   // row b0 - sgb
#pragma unroll
        for (int r = 0; r <= NR; ++r) {                        // rows b0-sgb, b0, ..., b0+(NR-1)sgb
            const bool inr = abs(b0 + (r - 1) * sgb) <= qm;
            vm[r] = buf_load_f64<(STREAM & 1) ? C2R_PLANE_AUX : 0>(r_prev, (inam && inr) ? o - da8 : kOOB);
            va_[r] = buf_load_f64<(STREAM & 1) ? C2R_PLANE_AUX : 0>(r_prev, (ina && inr) ? o : kOOB);
            o += db8;
        }
    }
    return shell_rows_fast_core<DET, LLS, STREAM, NR, EXT, STORE>(p, sa, ltab, thick, face, s, a, b0, sgb, nvalid, vm, va_, lds_cur);
}

#ifndef C2R_FAST_ATTR
#define C2R_FAST_ATTR __launch_bounds__(kBlock)
#endif
// one (source, face, tile) block of work of k_sweep_shell_fast; sl = position in the active list (src < 0: the source is
// sa.active[sl]; else src, and sl only indexes the loss partials of a shell on the sub-box surface)
template <bool DET, int LLS, int STREAM, int EXT, int LOOK = 0, int STORE = 1, int NR = kRows>
__device__ __forceinline__ void sweep_tile_fast(const KParams &p, const ShellArgs &sa, const FaceRect &fr, const v2f64 *ltab,
                                                const double *thick, double *sm, const int face, const int tile, const int sl,
                                                const ShellArgs &sq, const int src = -1)
{
    double loss = 0.0;
    const unsigned t = (unsigned)tile * kBlock + threadIdx.x;
    const unsigned bi = fr.magic ? __umulhi(t, fr.magic) : t;
    if (tile < fr.ntiles && bi < (unsigned)fr.npr) {
        const int a = fr.a_lo + (int)(t - __umul24(bi, (unsigned)fr.wa));
        const bool pos = bi < (unsigned)fr.pp;
        const int k0 = NR * (pos ? (int)bi : (int)bi - fr.pp);
        const int sgb = pos ? 1 : -1;
        const int b0 = pos ? k0 : -1 - k0;
        const int left = pos ? (fr.b_lo + fr.wb - b0) : (b0 - fr.b_lo + 1);
        loss = shell_rows_fast<DET, LLS, STREAM, NR, EXT, LOOK, STORE>(p, sa, ltab, thick, face, src < 0 ? sa.active[sl] : src, a, b0, sgb,
                                                                        min(left, NR), sq);
    }
    if (sa.has_boundary) {
        const double tot = block_sum_256(loss, sm);
        if (threadIdx.x == 0)
            sa.loss_partial[((size_t)sl * 6 + face) * sa.tiles_max + tile] = tot;
    }
}

template <bool DET, int LLS, bool STREAM, int EXT>
__global__ C2R_FAST_ATTR void k_sweep_shell_fast(KParams p, ShellArgs sa)
{
    __shared__ double sm[16];
    __shared__ v2f64 s_log[kBlock];                      // kBlock/64 waves x 64 entries
    const int face = blockIdx.y;
    const int tile = blockIdx.x;
    const int nact = *sa.n_active;
    if ((int)blockIdx.z >= nact) return;
    const FaceRect fr = sa.face[face];
    if (tile >= fr.ntiles && !sa.has_boundary) return;
    const v2f64 *ltab = wave_log_table(p.odtab, s_log);
    sweep_tile_fast<DET, LLS, STREAM, EXT>(p, sa, fr, ltab, p.thick, sm, face, tile, (int)blockIdx.z, sa);
}

// The same shell with an XCD-AWARE, MESH-PLANE-ORDERED block mapping (many sources, no cell on the sub-box surface).  Within a
// launch a mesh cell is visited by ~14 sources (q = 100, 1000 sources), but by workgroups that run far apart in time and on
// different XCDs, so n_HI (8 of the ~38 B per visit that leave the L2s) comes from HBM every time.  Workgroups are dealt
// round-robin over the 8 XCDs (blocks b and b + 8 share one: MI355X_MICROARCH.md, Workgroup dispatch -- a speed assumption only),
// each with its own 4 MB L2.  Here block b works for XCD group x = b % 8 on item b / 8 of that group's list: for every face, the
// x-th eighth of the batch's sources SORTED BY THEIR POSITION ALONG THE FACE'S AXIS (xa.perm, made on the host once per batch),
// source after source, tile after tile.  Sources whose faces lie on the same mesh plane are neighbours in that order: their
// tiles run on ONE XCD at about the same time, and the plane's n_HI is fetched from HBM once for all of them.  Same cells, same
// arithmetic, same results as k_sweep_shell / k_sweep_shell_fast (FAST); retired sources' blocks return at once (the host falls back to the compact
// active list when many have retired).
struct XcdArgs {
    const int *perm;           // [3][cap]: the batch's traceable sources sorted by srcw[axis] (axis 0, 1, 2), stable
    int n, cap;                // sources in each permutation; stride
    const int *final_nbox;     // [batch]: 0 while a source is being traced
};
template <bool DET, int LLS, bool STREAM, int EXT, bool FAST>
__global__ C2R_FAST_ATTR void k_sweep_shell_xcd(KParams p, ShellArgs sa, XcdArgs xa)
{
    __shared__ double sm[16];
    __shared__ v2f64 s_log[kBlock];
    const unsigned x = blockIdx.x & 7u;
    unsigned r = blockIdx.x >> 3;
    const unsigned start = (x * (unsigned)xa.n) >> 3, cnt = (((x + 1u) * (unsigned)xa.n) >> 3) - start;
    int face = -1; unsigned item = 0, tile = 0;
#pragma unroll
    for (int f = 0; f < 6; ++f) {
        const unsigned nt = (unsigned)sa.face[f].ntiles, nb = cnt * nt;
        if (face < 0) {
            if (r < nb) { face = f; item = r / nt; tile = r - item * nt; }
            else r -= nb;
        }
    }
    if (face < 0) return;
    const int s = xa.perm[(2 - (face >> 1)) * xa.cap + (int)(start + item)];
    if (xa.final_nbox[s] != 0) return;
    const FaceRect fr = sa.face[face];
    const v2f64 *ltab = wave_log_table(p.odtab, s_log);
    if (FAST) sweep_tile_fast<DET, LLS, STREAM ? 3 : 2, EXT>(p, sa, fr, ltab, p.thick, sm, face, (int)tile, 0, sa, s);
    else sweep_tile<DET, LLS, STREAM ? 3 : 2, EXT>(p, sa, fr, ltab, sm, face, (int)tile, 0, sa, s);
}

// Look-ahead pair (few sources, no cell of either shell on the sub-box surface): shell sa.q (blockIdx.y 0..5) and shell
// sb.q = sa.q + 1 (blockIdx.y 6..11) in ONE launch, both from the planes of shell sa.q - 1 -- the second shell recomputes
// the first shell's column densities where it needs them (lookahead_cd_out).  The first shell's planes are not stored
// (nobody reads them); the second shell's go to the other plane set, which the next launch reads.
template <bool DET, int LLS, bool STREAM, int EXT>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(1, 4)))     // latency-bound by design: registers before occupancy
void k_sweep_pair_fast(KParams p, ShellArgs sa, ShellArgs sb)
{
    __shared__ double sm[16];
    __shared__ v2f64 s_log[kBlock];                      // kBlock/64 waves x 64 entries
    const bool second = blockIdx.y >= 6;
    const int face = second ? (int)blockIdx.y - 6 : (int)blockIdx.y;
    const int tile = blockIdx.x;
    const int nact = *sa.n_active;
    if ((int)blockIdx.z >= nact) return;
    // the position table straight from global memory (it is a few L1 lines): filling the per-wave LDS copy is a memory round
    // trip in front of everything else, and a launch of this kernel is made of round trips
    const v2f64 *ltab = (C2R_PAIR_LDS_TABLE) ? wave_log_table(p.odtab, s_log) : p.odtab;
    if (second) {
        const FaceRect fr = sb.face[face];
        if (tile >= fr.ntiles) return;
        sweep_tile_fast<DET, LLS, STREAM, EXT, 1, true, kPairRows>(p, sb, fr, ltab, p.thick, sm, face, tile, (int)blockIdx.z, sa);
    } else {
        const FaceRect fr = sa.face[face];
        if (tile >= fr.ntiles) return;
        sweep_tile_fast<DET, LLS, STREAM, EXT, 0, false>(p, sa, fr, ltab, p.thick, sm, face, tile, (int)blockIdx.z, sa);
    }
}

// ---- the first sub-boxes, fused: one workgroup per source, all shells of the sub-box in one launch -----
// Near the source a shell has few cells (24q^2+2: 26 ... 602 for q = 1..5) and a launch per shell is
// nothing but latency, with the six faces' 256-thread tiles mostly empty.  Here the cells of a shell are
// packed over the six faces (face_off = prefix sums of the owned rectangles) and a source's workgroup walks
// the shells itself, a barrier between them; its photon loss through the box surface is summed in a fixed
// order and added to loss_acc[source] (no k_loss_reduce).  Same per-cell code as k_sweep_shell.
constexpr int kMaxFused = 5;
struct BoxArgs {
    int source_cell;                 // 1: the workgroup first does its source's own cell (sub-box 1; else k_source_cells has)
    int nshell;
    int ncell[kMaxFused];            // packed cells of each shell
    int face_off[kMaxFused][8];      // [f]: first packed index of face f; [6] = ncell
    ShellArgs sh[kMaxFused];
    const int *active;
    const int *n_active;
    double *loss_acc;
};

template <bool DET, int LLS, bool FAST, int EXT>
__global__ __launch_bounds__(1024) void k_sweep_box_fused(KParams p, BoxArgs ba)
{
    __shared__ double sm[16];
    __shared__ v2f64 s_log[1024];
    // FAST: the shells of this sub-box hand their column densities on through LDS -- two plane sets [6][kLdsP][kLdsP] (q <= 10: 21 KB
    // each), ping-ponged; only the launch's first shell reads the source's global planes and only its last writes them (the next
    // sub-box's launches read those).  The exact mode keeps the global planes (cell_state's buffer loads).
#ifdef C2R_FUSED_NO_LDS                  // (A/B builds: the shells of a fused sub-box hand over through the global planes, as before round 6)
    constexpr bool kLdsPlanes = false;
#else
    constexpr bool kLdsPlanes = FAST;
#endif
    __shared__ double s_planes[kLdsPlanes ? 2 * 6 * kLdsPP : 1];
    const int sl = blockIdx.x;
    if (sl >= *ba.n_active) return;
    const int s = ba.active[sl];
    const v2f64 *ltab = wave_log_table(p.odtab, s_log);       // both modes: the table positions of rates_fast
    if (ba.source_cell) {
        if (threadIdx.x == 0) {
            const ShellArgs &s0 = ba.sh[0];
            source_cell<EXT>(p, p.logtab, s, s0.boxR[0] == 0 || s0.boxR[1] == 0 || s0.boxR[2] == 0 || s0.boxL[0] == 0 ||
                              s0.boxL[1] == 0 || s0.boxL[2] == 0, ba.loss_acc, s0.dbg_cdout);
        }
        __syncthreads();             // shell 0's plane entries are visible to the waves that read them in shell 1
    }
    for (int k = 0; k < ba.nshell; ++k) {
        const ShellArgs &sa = ba.sh[k];
        double loss = 0.0;
        for (int t = threadIdx.x; t < ba.ncell[k]; t += blockDim.x) {
            int f = 0;
#pragma unroll
            for (int g = 1; g < 6; ++g) f += (t >= ba.face_off[k][g]) ? 1 : 0;
            const FaceRect fr = sa.face[f];
            const unsigned lt = (unsigned)(t - ba.face_off[k][f]);
            const unsigned bi = fr.magic ? __umulhi(lt, fr.magic) : lt;
            const int a = fr.a_lo + (int)(lt - __umul24(bi, (unsigned)fr.wa));
            const int b = fr.b_lo + (int)bi;
            if (FAST && !kLdsPlanes) loss = loss + shell_rows_fast<DET, LLS, false, 1, EXT>(p, sa, ltab, p.thick, f, s, a, b, b < 0 ? -1 : 1, 1, sa);
            else if (FAST) {
                // (block-uniform choices: the launch's first shell reads the global planes, its last one writes them)
                const double *lp = s_planes + ((k - 1) & 1) * 6 * kLdsPP;
                double *lc = s_planes + (k & 1) * 6 * kLdsPP;
                const int sg = b < 0 ? -1 : 1;
                if (k == 0) {
                    if (k + 1 < ba.nshell) loss = loss + shell_rows_fast<DET, LLS, false, 1, EXT, 0, 2>(p, sa, ltab, p.thick, f, s, a, b, sg, 1, sa, nullptr, lc);
                    else loss = loss + shell_rows_fast<DET, LLS, false, 1, EXT, 0, 1>(p, sa, ltab, p.thick, f, s, a, b, sg, 1, sa);
                } else {
                    if (k + 1 < ba.nshell) loss = loss + shell_rows_fast<DET, LLS, false, 1, EXT, 2, 2>(p, sa, ltab, p.thick, f, s, a, b, sg, 1, sa, lp, lc);
                    else loss = loss + shell_rows_fast<DET, LLS, false, 1, EXT, 2, 1>(p, sa, ltab, p.thick, f, s, a, b, sg, 1, sa, lp, nullptr);
                }
            } else loss = loss + shell_cell<DET, LLS, 0, EXT>(p, sa, ltab, f, s, a, b);
        }
        if (sa.has_boundary) {
            const double tot = block_sum_256(loss, sm);       // contains a barrier
            if (threadIdx.x == 0) ba.loss_acc[s] += tot;      // what k_loss_reduce does for the per-shell launches
        }
        __syncthreads();        // workgroup-scope release/acquire: this shell's planes are visible to every wave of the group
    }
}

// Deterministic mode: phih(cell) += Gamma_s(cell) for the sources of a batch IN SOURCE ORDER (the order
// of the serial reference, evolve_point.F90:283 inside master_slave.F90:85's loop).  One thread per
// cell; a source contributes where the cell lies inside its final sub-box (evolve_source.F90:135-136).
// A block is an 8 (x) by 32 (y) tile of one z-plane, a wave 8 by 8 cells: a source's rate sits in its x-fastest grid for cells
// of its z / y faces and in its y-fastest grid for cells of its x faces (that is how the sweep's waves wrote them), and
// groups of 8 consecutive cells in EITHER direction are one 64-byte sector -- both reads are sector-efficient.  (Round 5; the
// former row-of-256-cells mapping read the y-fastest grids with a stride of a mesh row: a sector per cell, 70 ms per pass at
// 256^3 x 1000 against ~35 now; the sums and their order are unchanged.)
__global__ __launch_bounds__(256) void k_gamma_reduce(KParams p, int nsrc, const int *final_nbox, int subbox,
                                                      double *phih, double *heat /* phiheat_grid, or null */,
                                                      const int *gate = nullptr /* see k_transpose_xy */)
{
    if (gate && *gate != 0) return;
    const int c0 = blockIdx.x * 8 + (threadIdx.x & 7), c1 = blockIdx.y * 32 + (threadIdx.x >> 3), c2 = blockIdx.z;
    const bool live = c0 < p.n[0] && c1 < p.n[1];
    const unsigned ncell = (unsigned)p.n[0] * (unsigned)p.n[1] * (unsigned)p.n[2];
    const unsigned id = (unsigned)c0 + (unsigned)p.n[0] * ((unsigned)c1 + (unsigned)p.n[1] * (unsigned)c2);
    const unsigned id_t = (unsigned)c1 + (unsigned)p.n[1] * ((unsigned)c0 + (unsigned)p.n[0] * (unsigned)c2);
    double acc = live ? phih[id] : 0.0;
    double acc_h = (live && heat) ? heat[id] : 0.0;
    for (int s = 0; s < nsrc; ++s) {
        const int nb = final_nbox[s];                 // uniform
        if (nb <= 0) continue;
        const int ext = subbox * nb;
        int d2 = c2 - p.srcw[3 * s + 2];              // uniform: the whole tile lies in one z-plane
        d2 -= (d2 > p.hr[2]) ? p.n[2] : 0;  d2 += (d2 < -p.hl[2]) ? p.n[2] : 0;
        if (d2 < -min(ext, p.hl[2]) || d2 > min(ext, p.hr[2])) continue;
        int d0 = c0 - p.srcw[3 * s + 0], d1 = c1 - p.srcw[3 * s + 1];
        d0 -= (d0 > p.hr[0]) ? p.n[0] : 0;  d0 += (d0 < -p.hl[0]) ? p.n[0] : 0;
        d1 -= (d1 > p.hr[1]) ? p.n[1] : 0;  d1 += (d1 < -p.hl[1]) ? p.n[1] : 0;
        if (live && d0 >= -min(ext, p.hl[0]) && d0 <= min(ext, p.hr[0]) && d1 >= -min(ext, p.hl[1]) && d1 <= min(ext, p.hr[1])) {
            const bool xf = abs(d0) > max(abs(d1), abs(d2));            // cinterp branch priority z > y > x
            const double *g = p.gbox + ((size_t)s * 2 + (xf ? 1 : 0)) * ncell;
            acc = acc + g[xf ? id_t : id];
            if (heat) acc_h = acc_h + (p.gbox_h + ((size_t)s * 2 + (xf ? 1 : 0)) * ncell)[xf ? id_t : id];
        }
    }
    if (live) phih[id] = acc;
    if (live && heat) heat[id] = acc_h;
}

// nhi[i,j,k] = max(1-max(xh_av,eps),eps) * ndens (ion%h_av(0)*ndens_p of evolve0D) and its (x,y)-transposed
// replica, one z-plane tile at a time through LDS.
// zero (null or 4 pointers): arrays of the mesh's size to clear on the way -- phih_grid and phiheat_grid (x fastest) in [0], [2],
// their (x,y)-transposed accumulators in [1], [3] (any of them null) -- where launches are what an iteration costs (c2r_iterate)
struct ZeroGrids { double *g[4]; };
// copy (or null pointers): n 32-bit words that block (0,0,0) copies on the way -- the pristine image of a small batch's state
// block (active lists, counters, accumulators) over the working one, instead of a host-to-device copy node in the graph
struct WordCopy { const unsigned *src; unsigned *dst; unsigned n; };
__global__ __launch_bounds__(256) void k_prepare_nhi(int n0, int n1, int n2, double eps, const float *__restrict__ ndens,
                                                     const double *__restrict__ xh_av, double *__restrict__ nhi,
                                                     double *__restrict__ nhi_T, ZeroGrids zero, WordCopy copy,
                                                     const double *__restrict__ xh_av0 = nullptr /* -DALLFRAC drivers: the stored neutral fraction */)
{
    __shared__ double tile[32][33];
    const int k = blockIdx.z;
    if (copy.n && blockIdx.x == 0 && blockIdx.y == 0 && k == 0)
        for (unsigned w = threadIdx.x; w < copy.n; w += 256) copy.dst[w] = copy.src[w];
    const int i0 = blockIdx.x * 32, j0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int i = i0 + tx, j = j0 + r;
        if (i < n0 && j < n1) {
            const size_t id = (size_t)i + (size_t)n0 * ((size_t)j + (size_t)n1 * k);
            const double xav1 = fmax(xh_av[id], eps);
            const double xav0 = xh_av0 ? fmax(xh_av0[id], eps)       // evolve_point.F90:131-132 (ALLFRAC)
                                       : fmax(1.0 - xav1, eps);      // :140
            const double v = xav0 * (double)ndens[id];
            nhi[id] = v;
            tile[r][tx] = v;
            if (zero.g[0]) zero.g[0][id] = 0.0;
            if (zero.g[2]) zero.g[2][id] = 0.0;
        }
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int j = j0 + tx, i = i0 + r;
        if (i < n0 && j < n1) {
            const size_t idt = (size_t)j + (size_t)n1 * ((size_t)i + (size_t)n0 * k);
            nhi_T[idt] = tile[tx][r];
            if (zero.g[1]) zero.g[1][idt] = 0.0;
            if (zero.g[3]) zero.g[3][idt] = 0.0;
        }
    }
}

// out[j + N1*(i + N0*k)] = in[i + N0*(j + N1*k)]: (x,y) transpose of every z-plane through LDS.
// gate (or null): the launch does nothing unless *gate == 0 -- kernels enqueued behind a sweep before the host knows
// whether every source has retired (the fused outer iteration, c2ray_hip.hip iterate_impl)
template <typename T, bool ACCUM>
__global__ __launch_bounds__(256) void k_transpose_xy(int n0, int n1, int n2, const T *__restrict__ in, T *__restrict__ out,
                                                      const int *gate = nullptr)
{
    __shared__ T tile[32][33];
    if (gate && *gate != 0) return;
    const int k = blockIdx.z;
    const int i0 = blockIdx.x * 32, j0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int i = i0 + tx, j = j0 + r;
        if (i < n0 && j < n1) tile[r][tx] = in[(size_t)i + (size_t)n0 * ((size_t)j + (size_t)n1 * k)];
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int j = j0 + tx, i = i0 + r;
        if (i < n0 && j < n1) {
            const size_t o = (size_t)j + (size_t)n1 * ((size_t)i + (size_t)n0 * k);
            if (ACCUM) out[o] += tile[tx][r]; else out[o] = tile[tx][r];
        }
    }
}

// dst += src over the mesh (the two reduced halves of an overlapped pass: sweep.hip overlap_end)
__global__ __launch_bounds__(256) void k_add_grid(size_t n, const double *__restrict__ src, double *__restrict__ dst)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = dst[i] + src[i];
}

// ---- evolve0D for ONE (cell, source): the reference's per-cell call surface (evolve_point.F90:83-299) ----------------
// Slow by construction (a launch and a handful of copies per cell): for hosts that drive the sweep themselves and for tests.
// The cell's geometry as the shell kernels see it (face, plane coordinates a, b, shell sa.q), the four upstream column
// densities as VALUES (the caller reads them from its coldensh_out), n_HI from the context's arrays; the arithmetic is
// cell_state's (bit-identical column densities) and rates_fast's, the source cell's is source_cell's.
// out[0..3] = coldensh_out(pos), the rate to add to phih_grid(pos), to phiheat_grid(pos), the photon loss through the box surface
template <int LLS, int EXT>
__global__ __launch_bounds__(64) void k_evolve0d_cell(KParams p, ShellArgs sa, int face, int a, int b, int is_source, int on_surface,
                                                      double c1, double c2, double c3, double c4, double *out)
{
    __shared__ v2f64 s_tab[kLogTab];
    const v2f64 *ltab = wave_log_table(is_source ? p.logtab : p.odtab, s_tab);
    if (threadIdx.x != 0) return;
    const int s = 0;
    const double nflux = p.normflux[s];
    double cd_out, gamma = 0.0, heat = 0.0, loss = 0.0;
    if (is_source) {                                             // evolve_point.F90:151-160
        const unsigned i = wrap_pos(p.srcw[0], p.n[0], 0), j = wrap_pos(p.srcw[1], p.n[1], 0), k = wrap_pos(p.srcw[2], p.n[2], 0);
        const double nhi = p.nhi[(size_t)i + (size_t)p.n[0] * ((size_t)j + (size_t)p.n[1] * (size_t)k)];
        const double path = 0.5 * step_of(p).dr[0];
        const double vol_ph = step_of(p).dr[0] * step_of(p).dr[1] * step_of(p).dr[2];
        cd_out = 0.0 + nhi * path;
        if (nflux > 0.0) {
            double p_out = 0.0;
            gamma = photoion<EXT>(p, ltab, 0.0, cd_out, vol_ph, nflux, p_out, &heat, (EXT & 2) ? p.normflux_x[s] : 0.0) / nhi;
            if (on_surface) loss = p_out * step_of(p).vol / vol_ph;
        }
    } else {
        const CellState cs = cell_state<LLS, false>(p, sa, face, s, a, b, c1, c2, c3, c4, weight_rcp(p, c1), weight_rcp(p, c2),
                                                    weight_rcp(p, c3), weight_rcp(p, c4));
        cd_out = cs.cd_out;
        if (!cs.stop_far && !(cs.cd_in > p.max_coldensh) && nflux > 0.0) {
            double p_out;
            gamma = rates_fast<EXT>(p, ltab, p.thick, cs.cd_in, cd_out, nflux, (EXT & 2) ? p.normflux_x[s] : 0.0, cs.vol_ph * cs.nhi, cs.vol_ph, p_out, heat);
            if (on_surface) loss = fdiv(p_out * step_of(p).vol, cs.vol_ph);
        }
    }
    out[0] = cd_out; out[1] = gamma; out[2] = heat; out[3] = loss;
}

// Adds the block partials of one shell launch to loss_acc[source], in a fixed order.
__global__ __launch_bounds__(256) void k_loss_reduce(const int *active, const int *n_active, const double *loss_partial,
                                                     int bps, double *loss_acc)
{
    __shared__ double sm[4];
    const int sl = blockIdx.x;
    if (sl >= *n_active) return;
    double v = 0.0;
    for (int i = threadIdx.x; i < bps; i += 256) v += loss_partial[(size_t)sl * bps + i];   // bps = 6*tiles_max
    const double tot = block_sum_256(v, sm);
    if (threadIdx.x == 0) loss_acc[active[sl]] += tot;
}

// End of sub-box `nbox` (evolve_source.F90:128-131): keep a source active while more than
// loss_fraction of its photons leave the box and the box can still grow in z.  Compacts the
// active list (stable), finalises the others.  One block of 1024 threads.
__global__ __launch_bounds__(1024) void k_box_decide(const int *active_in, const int *n_in_dev, int *active_out,
                                                     int *n_out, int *n_out_host, const double *normflux, double S_star,
                                                     double loss_fraction, int can_grow, int nbox,
                                                     double *loss_acc, double *final_loss, int *final_nbox,
                                                     const double *loss_partial, int bps, int next_bound, int *halt_host)
{
    // next_bound / halt_host (a replayed launch sequence, sweep.hip run_chains): the launches of the next sub-box were sized for
    // next_bound sources when they were captured.  Should more stay active, the device count is left at ZERO -- every later launch
    // of the sequence returns at once, the lists and planes stay as this sub-box left them -- the true count still goes to the
    // host's slot and the sub-box number to *halt_host: the host resumes from there with launches of the right size.
    // loss_partial/bps: block partials of the sub-box's LAST shell launch (bps = 6 x tiles per source, 0: none) -- what
    // k_loss_reduce would add, folded in here to save a launch; four interleaved partial sums, then in a fixed order
    __shared__ int scan[1024];
    __shared__ int base;
    const int n_in = *n_in_dev;
    if (threadIdx.x == 0) base = 0;
    __syncthreads();
    for (int start = 0; start < n_in; start += 1024) {
        const int i = start + (int)threadIdx.x;
        int keep = 0, s = -1;
        if (i < n_in) {
            s = active_in[i];
            const double flux = normflux[s] * S_star;
            double loss = loss_acc[s];
            if (bps > 0) {
                const double *pp = loss_partial + (size_t)i * bps;
                double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
                int j = 0;
                for (; j + 3 < bps; j += 4) { a0 += pp[j]; a1 += pp[j + 1]; a2 += pp[j + 2]; a3 += pp[j + 3]; }
                for (; j < bps; ++j) a0 += pp[j];
                loss = loss + ((a0 + a1) + (a2 + a3));
                loss_acc[s] = loss;
            }
            keep = (loss > loss_fraction * flux) && can_grow;
            if (keep) loss_acc[s] = 0.0;                      // evolve_source.F90:133
            else { final_loss[s] = loss; final_nbox[s] = nbox; }
        }
        scan[threadIdx.x] = keep;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {            // inclusive Hillis-Steele scan
            int v = 0;
            if ((int)threadIdx.x >= off) v = scan[threadIdx.x - off];
            __syncthreads();
            scan[threadIdx.x] += v;
            __syncthreads();
        }
        if (keep) active_out[base + scan[threadIdx.x] - 1] = s;
        __syncthreads();
        if (threadIdx.x == 1023) base += scan[1023];
        __syncthreads();
    }
    // n_out_host: the schedule's pinned slot for this sub-box, written straight through the mapped pointer
    if (threadIdx.x == 0) {
        const bool over = base > next_bound;
        *n_out = over ? 0 : base; *n_out_host = base;
        if (over) *halt_host = nbox;
    }
}

// k_box_decide for up to 64 active sources (one wave, no scan through LDS): the same sums in the same order, the same
// decisions.  tot.on: when the decision leaves no source active, the batch's totals (k_batch_totals for a pass of ONE batch:
// the running sums restart from zero) and the per-source results go out with it -- to the device scalars and, through
// mapped pointers, to the host -- so that the launches behind a small batch need neither k_batch_totals nor two copies.
struct SmallTotals {
    int on, nsrc;
    double *photon_loss; long long *sum_nbox;         // device running totals
    double *host_loss; long long *host_nbox;          // the host's pinned scalars
    int *host_final_nbox; double *host_final_loss;    // the host's pinned staging arrays [nsrc]
};
__global__ __launch_bounds__(64) void k_box_decide_small(const int *active_in, const int *n_in_dev, int *active_out,
                                                         int *n_out, int *n_out_host, const double *normflux, double S_star,
                                                         double loss_fraction, int can_grow, int nbox,
                                                         double *loss_acc, double *final_loss, int *final_nbox,
                                                         const double *loss_partial, int bps, SmallTotals tot,
                                                         int next_bound, int *halt_host)
{
#if defined(__AMDGCN_WAVEFRONT_SIZE) && __AMDGCN_WAVEFRONT_SIZE != 64
#error "k_box_decide_small is one 64-lane wave (ballot, cross-lane reads of final_loss): build for a wave64 target (gfx950)"
#endif
    const int n_in = *n_in_dev;                       // <= 64
    const int i = threadIdx.x;
    int keep = 0, s = -1;
    if (i < n_in) {
        s = active_in[i];
        const double flux = normflux[s] * S_star;
        double loss = loss_acc[s];
        if (bps > 0) {
            const double *pp = loss_partial + (size_t)i * bps;
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
            int j = 0;
            for (; j + 3 < bps; j += 4) { a0 += pp[j]; a1 += pp[j + 1]; a2 += pp[j + 2]; a3 += pp[j + 3]; }
            for (; j < bps; ++j) a0 += pp[j];
            loss = loss + ((a0 + a1) + (a2 + a3));
            loss_acc[s] = loss;
        }
        keep = (loss > loss_fraction * flux) && can_grow;
        if (keep) loss_acc[s] = 0.0;                      // evolve_source.F90:133
        else { final_loss[s] = loss; final_nbox[s] = nbox; }
    }
    const unsigned long long mask = __ballot(keep);
    if (keep) active_out[__popcll(mask & ((1ULL << i) - 1ULL))] = s;
    const int n_keep = __popcll(mask);
    if (i == 0) {                                         // (next_bound, halt_host: see k_box_decide)
        const bool over = n_keep > next_bound;
        *n_out = over ? 0 : n_keep; *n_out_host = n_keep;
        if (over) *halt_host = nbox;
    }
    if (tot.on && n_keep == 0) {
        __threadfence_block();                            // (one wave: the stores above are ordered before the loads below)
        __syncthreads();
        for (int t = i; t < tot.nsrc; t += 64) { tot.host_final_nbox[t] = final_nbox[t]; tot.host_final_loss[t] = final_loss[t]; }
        if (i == 0) {
            double L = 0.0; long long NB = 0;
            for (int t = 0; t < tot.nsrc; ++t) { L = L + (0.0 + final_loss[t]); NB += final_nbox[t]; }
            *tot.photon_loss = L; *tot.sum_nbox = NB;
            *tot.host_loss = L; *tot.host_nbox = NB;
        }
    }
}

// Behind the replayed launch sequences of a pass's chains (sweep.hip run_chains): *gate = 0 iff every chain's sequence ran to its
// end with no source left to trace and none was halted -- what the launches behind the pass (totals, the fold of the transposed
// rates, the global pass) wait for on the device instead of the host.
constexpr int kGateChains = 4;
struct ChainGateArgs { const int *n_active[kGateChains]; const int *halt[kGateChains]; int nch; };
__global__ void k_chain_gate(ChainGateArgs a, int *gate)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int g = 0;
    for (int c = 0; c < a.nch; ++c) g += *a.n_active[c] + (*a.halt[c] != 0 ? 1 : 0);
    *gate = g;
}

// photon_loss(1) += photon_loss_src, in source order (evolve_source.F90:216); sum_nbox (:219).
// first: first batch of a pass (the running totals restart from zero).  The totals so far are also
// written to the host's pinned scalars through their mapped pointers.
__global__ __launch_bounds__(1024) void k_batch_totals(int nsrc, const double *final_loss, const int *final_nbox,
                                                       double *photon_loss, long long *sum_nbox, int first,
                                                       double *host_loss, long long *host_nbox, const int *gate = nullptr /* see k_transpose_xy */)
{
    if (gate && *gate != 0) return;
    // thread t sums the contiguous run [t*c, (t+1)*c) in order, thread 0 the runs in order: for up to 1024
    // sources (c = 1) that IS the sequential source-order sum of the reference; beyond, a fixed two-level order
    __shared__ double sl[1024];
    __shared__ long long sn[1024];
    const int c = (nsrc + 1023) / 1024;
    const int i0 = (int)threadIdx.x * c, i1 = min(nsrc, i0 + c);
    double l = 0.0; long long nb = 0;
    for (int i = i0; i < i1; ++i) { l = l + final_loss[i]; nb += final_nbox[i]; }
    sl[threadIdx.x] = l; sn[threadIdx.x] = nb;
    __syncthreads();
    if (threadIdx.x == 0) {
        double L = first ? 0.0 : *photon_loss; long long NB = first ? 0 : *sum_nbox;
        const int used = c > 0 ? (nsrc + c - 1) / c : 0;
        for (int t = 0; t < used; ++t) { L = L + sl[t]; NB += sn[t]; }
        *photon_loss = L; *sum_nbox = NB;
        *host_loss = L; *host_nbox = NB;
    }
}

}  // namespace c2r
