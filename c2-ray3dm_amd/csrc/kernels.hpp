// Device kernels of the C2-Ray evolve hot path for gfx950 (MI355X, CDNA4).
//
// Design:
//  * The short-characteristics sweep of one source is causal only from one Chebyshev shell
//    (cube surface |d|_inf = q) to the next: every upstream cell that cinterp gives a non-zero
//    weight lies in shell q-1 (column_density.f90:108,173,226).  So shell q of ALL sources of a
//    batch is one launch; its 24q^2+2 cells per source are independent.
//  * A source's column densities live only in two "shell" buffers (planes of the 6 cube faces,
//    ping-pong by q parity), not in an N^3 array per source (evolve_data.F90 coldensh_out).
//  * f64 throughout, -ffp-contract=off: statement order follows the reference so that results
//    agree with the Fortran to rounding of the transcendental functions only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace c2r {

constexpr int kTileA = 64;   // lanes of a wave run along the fast plane axis
constexpr int kTileB = 4;    // one wave per plane row, 4 rows per 256-thread block
constexpr int kBlock = kTileA * kTileB;

struct KParams {
    int n[3];
    int hl[3], hr[3];          // trace limits around a source: -hl..+hr (evolve_source.F90:100-102)
    double dr[3];
    double vol;
    double coldensh_LLS;
    double sigma, wfloor, sqrt2, sqrt3, fourpi;
    double max_coldensh, tau_limit, minlogtau, dlogtau, numtau_d, eps;
    int numtau;
    int R, P;                  // plane centre offset and pitch (P = 2R+1)
    size_t PP;                 // P*P
    const float  *ndens;
    const double *xh_av;
    double *phih;
    const double *thick, *thin;
    const int    *srcpos;      // 3 x S_batch (unwrapped, 1-based)
    const double *normflux;    // S_batch
    double *planes;            // [S_batch][2][6][P][P]
};

struct ShellArgs {
    int q;
    int tiles_a, tiles_b, bps;   // tiles per face plane, blocks per source
    int boxR[3], boxL[3];        // limits of the current sub-box (last_r/last_l - srcpos)
    int has_boundary;
    double alam;                 // (q-0.5)/q, column_density.f90:112 (sign cancels)
    const int *active;           // compacted list of local source indices
    double *loss_partial;        // [n_active][bps]
    double *dbg_cdout;           // optional N^3 coldensh_out of the (single) source, else null
};

__device__ __forceinline__ int pmod(int a, int n) { int r = a % n; return r < 0 ? r + n : r; }

// radiation_photoionrates.F90:184-228  set_tau_table_positions + read_table
__device__ __forceinline__ double table_lookup(const double *__restrict__ tab, double tau,
                                               const KParams &p)
{
    const double lt = log10(fmax(1.0e-20, tau));
    const double od = fmin(p.numtau_d, fmax(0.0, 1.0 + (lt - p.minlogtau) / p.dlogtau));
    const int ip = (int)od;
    const double res = od - (double)ip;
    const int ip1 = min(p.numtau, ip + 1);
    const double t0 = tab[ip], t1 = tab[ip1];
    return t0 + (t1 - t0) * res;
}

// radiation_photoionrates.F90:71-179, :233-317 for NumFreqBnd=1, stellar table.
// Returns photo_cell_HI (already divided by vol_ph); out = photo_out.
__device__ __forceinline__ double photoion(const KParams &p, double cd_in, double cd_out,
                                           double vol_ph, double nflux, double &p_out)
{
    const double tau_in = cd_in * p.sigma, tau_out = cd_out * p.sigma;
    const double p_in = nflux * table_lookup(p.thick, tau_in, p);
    double p_cell;
    if (fabs(tau_out - tau_in) > p.tau_limit) {
        p_out = nflux * table_lookup(p.thick, tau_out, p);
        p_cell = p_in - p_out;
    } else {
        p_cell = nflux * (tau_out - tau_in) * table_lookup(p.thin, tau_in, p);
        p_out = p_in - p_cell;
    }
    return p_cell / vol_ph;
}

// Deterministic block sum (fixed order): wave shuffles, then the 4 wave sums in order.
__device__ __forceinline__ double block_sum_256(double v, double *sm /* >= 4 doubles */)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int tid = threadIdx.y * blockDim.x + threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    if (lane == 0) sm[wave] = v;
    __syncthreads();
    double r = 0.0;
    if (tid == 0) { const int nw = (blockDim.x * blockDim.y) >> 6; for (int w = 0; w < nw; ++w) r += sm[w]; }
    return r;   // valid in thread 0
}

// ---- source cells (q = 0) ------------------------------------------------------------------
// evolve_point.F90:151-160 (source cell) + the common tail of evolve0D.  One thread per source.
__global__ void k_source_cells(KParams p, int nsrc, const int *active, int boxR0, int boxR1, int boxR2,
                               int boxL0, int boxL1, int boxL2, double *loss_acc, double *dbg_cdout)
{
    const int sl = blockIdx.x * blockDim.x + threadIdx.x;
    if (sl >= nsrc) return;
    const int s = active[sl];
    const int i = pmod(p.srcpos[3 * s + 0] - 1, p.n[0]);
    const int j = pmod(p.srcpos[3 * s + 1] - 1, p.n[1]);
    const int k = pmod(p.srcpos[3 * s + 2] - 1, p.n[2]);
    const size_t id = (size_t)i + (size_t)p.n[0] * ((size_t)j + (size_t)p.n[1] * (size_t)k);
    const double xav1 = fmax(p.xh_av[id], p.eps);
    const double xav0 = fmax(1.0 - xav1, p.eps);
    const double nd = (double)p.ndens[id];
    const double cd_in = 0.0;
    const double path = 0.5 * p.dr[0];
    const double vol_ph = p.dr[0] * p.dr[1] * p.dr[2];
    const double cd_out = cd_in + xav0 * nd * path;
    // plane q=0 of every face is the single cell (0,0)
    for (int f = 0; f < 6; ++f)
        p.planes[((size_t)s * 2 + 0) * 6 * p.PP + (size_t)f * p.PP + (size_t)p.R * p.P + p.R] = cd_out;
    if (dbg_cdout) dbg_cdout[id] = cd_out;
    const double nflux = p.normflux[s];
    double p_out = 0.0, gamma = 0.0;
    if (nflux > 0.0) {      // cd_in = 0 is never above max_coldensh
        gamma = photoion(p, cd_in, cd_out, vol_ph, nflux, p_out) / (xav0 * nd);
        atomicAdd(&p.phih[id], gamma);
    }
    // degenerate meshes only: the source cell itself sits on the sub-box surface
    if (boxR0 == 0 || boxR1 == 0 || boxR2 == 0 || boxL0 == 0 || boxL1 == 0 || boxL2 == 0)
        loss_acc[s] += p_out * p.vol / vol_ph;
}

// ---- one Chebyshev shell of every active source ------------------------------------------------
// evolve0D (evolve_point.F90:83-299) + cinterp (column_density.f90:29-271) + photoion_rates.
// faces: 0:+z 1:-z 2:+y 3:-y 4:+x 5:-x.  Plane coordinates (a,b): z-face (x,y); y-face (x,z);
// x-face (y,z).  A cell on an edge/corner of the cube belongs to the face of highest priority
// (z over y over x: the branch order of cinterp); its owner also stores it into the other
// faces' planes, which read it in shell q+1.
__global__ __launch_bounds__(kBlock) void k_sweep_shell(KParams p, ShellArgs sa)
{
    __shared__ double sm[4];
    const int blk = blockIdx.x;
    const int sl = blk / sa.bps;
    int r = blk - sl * sa.bps;
    const int s = sa.active[sl];
    const int tiles = sa.tiles_a * sa.tiles_b;
    const int face = r / tiles;  r -= face * tiles;
    const int tb = r / sa.tiles_a, ta = r - tb * sa.tiles_a;
    const int q = sa.q;
    const int axis = 2 - (face >> 1);            // 2:z 1:y 0:x
    const int sg = (face & 1) ? -1 : 1;
    const int pd = sg * q;
    const int a = -q + ta * kTileA + (int)threadIdx.x;
    const int b = -q + tb * kTileB + (int)threadIdx.y;

    // mesh-axis deltas of this cell (selects, not a runtime-indexed array: that would spill)
    const int d0 = (axis == 0) ? pd : a;
    const int d1 = (axis == 2) ? b : ((axis == 1) ? pd : a);
    const int d2 = (axis == 2) ? pd : b;
    bool valid = (abs(a) <= q) && (abs(b) <= q);
    valid = valid && d0 >= -p.hl[0] && d0 <= p.hr[0] && d1 >= -p.hl[1] && d1 <= p.hr[1] &&
            d2 >= -p.hl[2] && d2 <= p.hr[2];
    const bool owner = (axis == 2) || (axis == 1 && abs(b) < q) || (axis == 0 && abs(a) < q && abs(b) < q);

    double loss = 0.0;
    if (valid && owner) {
        const int s0 = p.srcpos[3 * s + 0], s1 = p.srcpos[3 * s + 1], s2 = p.srcpos[3 * s + 2];
        const int su = (axis == 0) ? s1 : s0;      // source coordinate on plane axis a / b
        const int sv = (axis == 2) ? s1 : s2;
        const int i = pmod(s0 + d0 - 1, p.n[0]);
        const int j = pmod(s1 + d1 - 1, p.n[1]);
        const int k = pmod(s2 + d2 - 1, p.n[2]);
        const size_t id = (size_t)i + (size_t)p.n[0] * ((size_t)j + (size_t)p.n[1] * (size_t)k);
        const double xav_raw = p.xh_av[id];
        const double nd = (double)p.ndens[id];

        // upstream corners in plane q-1 of this face (zero weight outside |.| <= q-1)
        const int sga = a < 0 ? -1 : 1, sgb = b < 0 ? -1 : 1;
        const int am = a - sga, bm = b - sgb;
        const double *prev = p.planes + ((size_t)s * 2 + ((q - 1) & 1)) * 6 * p.PP + (size_t)face * p.PP;
        const int qm = q - 1;
        const bool ina = abs(a) <= qm, inam = abs(am) <= qm, inb = abs(b) <= qm, inbm = abs(bm) <= qm;
        const double c1 = (inam && inbm) ? prev[(size_t)(bm + p.R) * p.P + (am + p.R)] : 0.0;
        const double c2 = (ina && inbm) ? prev[(size_t)(bm + p.R) * p.P + (a + p.R)] : 0.0;
        const double c3 = (inam && inb) ? prev[(size_t)(b + p.R) * p.P + (am + p.R)] : 0.0;
        const double c4 = (ina && inb) ? prev[(size_t)(b + p.R) * p.P + (a + p.R)] : 0.0;

        // cinterp, generic in (a,b,pd): the three branches differ only by which axes play (u,v)
        const double du = (double)(float)a, dv = (double)(float)b, dp = (double)(float)pd;
        const double uc = sa.alam * du + (double)(float)su;
        const double vc = sa.alam * dv + (double)(float)sv;
        const double ddu = 2.0 * fabs(uc - (double)((float)(su + am) + 0.5f * (float)sga));
        const double ddv = 2.0 * fabs(vc - (double)((float)(sv + bm) + 0.5f * (float)sgb));
        const double w1 = ((1. - ddu) * (1. - ddv)) * (1.0 / fmax(p.wfloor, c1 * p.sigma));
        const double w2 = ((1. - ddv) * ddu) * (1.0 / fmax(p.wfloor, c2 * p.sigma));
        const double w3 = ((1. - ddu) * ddv) * (1.0 / fmax(p.wfloor, c3 * p.sigma));
        const double w4 = (ddu * ddv) * (1.0 / fmax(p.wfloor, c4 * p.sigma));
        double cdi = (c1 * w1 + c2 * w2 + c3 * w3 + c4 * w4) / (w1 + w2 + w3 + w4);
        if (q == 1 && (abs(a) == 1 || abs(b) == 1))
            cdi = (abs(a) == 1 && abs(b) == 1) ? p.sqrt3 * cdi : p.sqrt2 * cdi;
        double path = sqrt((du * du + dv * dv) / (dp * dp) + 1.0);

        // evolve0D
        path = path * p.dr[0];
        const double xs = p.dr[0] * (double)(float)d0;
        const double ys = p.dr[1] * (double)(float)d1;
        const double zs = p.dr[2] * (double)(float)d2;
        const double dist2 = xs * xs + ys * ys + zs * zs;
        const double vol_ph = p.fourpi * dist2 * path;
        const double cd_in = cdi + p.coldensh_LLS * path / p.dr[0];
        const double xav1 = fmax(xav_raw, p.eps);
        const double xav0 = fmax(1.0 - xav1, p.eps);
        const double cd_out = cd_in + xav0 * nd * path;

        // store into this face's plane and into the planes of the faces sharing the cell
        double *cur = p.planes + ((size_t)s * 2 + (q & 1)) * 6 * p.PP;
        cur[(size_t)face * p.PP + (size_t)(b + p.R) * p.P + (a + p.R)] = cd_out;
        if (axis == 2) {
            if (abs(a) == q)   // x-face (u=y=b, v=z=pd)
                cur[(size_t)(a > 0 ? 4 : 5) * p.PP + (size_t)(pd + p.R) * p.P + (b + p.R)] = cd_out;
            if (abs(b) == q)   // y-face (u=x=a, v=z=pd)
                cur[(size_t)(b > 0 ? 2 : 3) * p.PP + (size_t)(pd + p.R) * p.P + (a + p.R)] = cd_out;
        } else if (axis == 1) {
            if (abs(a) == q)   // x-face (u=y=pd, v=z=b)
                cur[(size_t)(a > 0 ? 4 : 5) * p.PP + (size_t)(b + p.R) * p.P + (pd + p.R)] = cd_out;
        }
        if (sa.dbg_cdout) sa.dbg_cdout[id] = cd_out;

        const double nflux = p.normflux[s];
        if (!(cd_in > p.max_coldensh) && nflux > 0.0) {
            double p_out;
            const double gamma = photoion(p, cd_in, cd_out, vol_ph, nflux, p_out) / (xav0 * nd);
            if (gamma != 0.0) atomicAdd(&p.phih[id], gamma);
            if (sa.has_boundary) {
                const bool bnd = d0 == sa.boxR[0] || d1 == sa.boxR[1] || d2 == sa.boxR[2] ||
                                 d0 == -sa.boxL[0] || d1 == -sa.boxL[1] || d2 == -sa.boxL[2];
                if (bnd) loss = p_out * p.vol / vol_ph;
            }
        }
    }
    if (sa.has_boundary) {
        const double tot = block_sum_256(loss, sm);
        if (threadIdx.x == 0 && threadIdx.y == 0) sa.loss_partial[(size_t)sl * sa.bps + (blk - sl * sa.bps)] = tot;
    }
}

// Adds the block partials of one shell launch to loss_acc[source], in a fixed order.
__global__ __launch_bounds__(256) void k_loss_reduce(const int *active, const double *loss_partial, int bps,
                                                     double *loss_acc)
{
    __shared__ double sm[4];
    const int sl = blockIdx.x;
    double v = 0.0;
    for (int i = threadIdx.x; i < bps; i += 256) v += loss_partial[(size_t)sl * bps + i];
    const double tot = block_sum_256(v, sm);
    if (threadIdx.x == 0) loss_acc[active[sl]] += tot;
}

// End of sub-box `nbox` (evolve_source.F90:128-131): keep a source active while more than
// loss_fraction of its photons leave the box and the box can still grow in z.  Compacts the
// active list (stable), finalises the others.  One block of 1024 threads.
__global__ __launch_bounds__(1024) void k_box_decide(const int *active_in, int n_in, int *active_out,
                                                     int *n_out, const double *normflux, double S_star,
                                                     double loss_fraction, int can_grow, int nbox,
                                                     double *loss_acc, double *final_loss, int *final_nbox)
{
    __shared__ int scan[1024];
    __shared__ int base;
    if (threadIdx.x == 0) base = 0;
    __syncthreads();
    for (int start = 0; start < n_in; start += 1024) {
        const int i = start + (int)threadIdx.x;
        int keep = 0, s = -1;
        if (i < n_in) {
            s = active_in[i];
            const double flux = normflux[s] * S_star;
            const double loss = loss_acc[s];
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
    if (threadIdx.x == 0) *n_out = base;
}

// photon_loss(1) += photon_loss_src, in source order (evolve_source.F90:216); sum_nbox (:219).
__global__ void k_batch_totals(int nsrc, const double *final_loss, const int *final_nbox,
                               double *photon_loss, long long *sum_nbox)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        double l = *photon_loss; long long nb = *sum_nbox;
        for (int s = 0; s < nsrc; ++s) { l = l + final_loss[s]; nb += final_nbox[s]; }
        *photon_loss = l; *sum_nbox = nb;
    }
}

// ---- global pass -------------------------------------------------------------------------------
struct ChemParams {
    double dt, eps, min_frac_change, min_frac_atoms, abu_c, deltht_small;
    double brech0, acolh0;        // doric.f90:73,78 evaluated on the host for the step's temperature
    int max_iter;
};

// evolve0D_global (evolve_point.F90:305-406) + do_chemistry (:410-555) + doric (doric.f90:33-134).
// Fixed grid, grid-stride: block partial sums of xh_intermed land in sum_partial[blockIdx.x].
__global__ __launch_bounds__(256) void k_global_pass(ChemParams c, size_t ncell, const float *__restrict__ ndens,
                                                     const double *__restrict__ xh, double *__restrict__ xh_av,
                                                     double *__restrict__ xh_intermed,
                                                     const double *__restrict__ phih, double *sum_partial,
                                                     unsigned long long *conv_flag, unsigned int *chem_fail)
{
    __shared__ double sm[4];
    double lsum = 0.0;
    unsigned int nconv = 0, nfail = 0;
    for (size_t id = (size_t)blockIdx.x * 256 + threadIdx.x; id < ncell; id += (size_t)gridDim.x * 256) {
        const double h_old1 = fmax(c.eps, xh[id]);
        const double xav_in = xh_av[id];
        double hav1 = fmax(c.eps, xav_in);
        const double h_old0 = 1.0 - h_old1;
        double hav0 = 1.0 - hav1;
        const double nd = (double)ndens[id];
        const double gamma = phih[id];
        double h1 = h_old1, h0 = h_old0;
        int nit = 0;
        for (;;) {
            nit++;
            const double yh0_av_old = hav0;
            const double de = nd * (hav1 + c.abu_c);                 // tped.f90:81
            const double aih0 = gamma + de * c.acolh0;
            const double delth = aih0 + de * c.brech0;
            const double eq1 = aih0 / delth;
            const double eq0 = de * c.brech0 / delth;
            const double deltht = delth * c.dt;
            const double ee = exp(-deltht);
            h1 = (h_old1 - eq1) * ee + eq1;
            h0 = (h_old0 - eq0) * ee + eq0;
            if (h0 < c.eps) { h0 = c.eps; h1 = 1.0 - c.eps; }
            const double avg = deltht < c.deltht_small ? 1.0 : (1.0 - ee) / deltht;
            hav1 = eq1 + (h_old1 - eq1) * avg;
            hav0 = 1.0 - hav1;
            if (hav0 < c.eps) hav0 = c.eps;
            if (fabs((hav0 - yh0_av_old) / hav0) < c.min_frac_change || hav0 < c.min_frac_atoms) break;
            if (nit > c.max_iter) { nfail++; break; }
        }
        const double yh0_old = 1.0 - fmax(c.eps, xav_in);           // evolve_point.F90:378-379
        if (fabs(hav0 - yh0_old) > c.min_frac_change && fabs((hav0 - yh0_old) / hav0) > c.min_frac_change &&
            hav0 > c.min_frac_atoms) nconv++;
        xh_intermed[id] = h1;
        xh_av[id] = hav1;
        lsum += h1;
    }
    const double tot = block_sum_256(lsum, sm);
    if (threadIdx.x == 0) sum_partial[blockIdx.x] = tot;
    // integer counts: order-independent
    for (int off = 32; off > 0; off >>= 1) { nconv += __shfl_down(nconv, off, 64); nfail += __shfl_down(nfail, off, 64); }
    if ((threadIdx.x & 63) == 0) {
        if (nconv) atomicAdd(conv_flag, (unsigned long long)nconv);
        if (nfail) atomicAdd(chem_fail, nfail);
    }
}

__global__ __launch_bounds__(256) void k_sum_partial(size_t n, const double *__restrict__ a, double *partial)
{
    __shared__ double sm[4];
    double v = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) v += a[i];
    const double tot = block_sum_256(v, sm);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void k_sum_final(int n, const double *partial, double *out)
{
    __shared__ double sm[4];
    double v = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) v += partial[i];
    const double tot = block_sum_256(v, sm);
    if (threadIdx.x == 0) *out = tot;
}

}  // namespace c2r
