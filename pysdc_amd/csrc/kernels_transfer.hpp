// libsdcmi kernels: space transfer between grids.
#pragma once
#include "context.hpp"

// ------------------------------------------------------------------------------------------------------
// space transfer between nested periodic grids (coarsening by 2 per axis): tensor products of the 1-D
// interpolation  fine[2i] = coarse[i], fine[2i+1] = sum_j w[j] coarse[i - k/2 + 1 + j]  and of its scaled
// transpose (TransferMesh.py:49-146 with helpers/transfer_helper.py:153-186, periodic / equidist_nested)
// ------------------------------------------------------------------------------------------------------
struct XferArgs {
    const double* in;
    double* out;
    const int* idx;     // [W][n_out] source indices along the axis (device), entry j of output row i at j * n_out + i
    const double* w;    // [W][n_out] weights (zero-padded)
    size_t outer, inner;
    int n_out, n_in, W;
    int accumulate;     // 1: out += result (the coarse-grid correction added to the node values by the pass that makes it)
};

// one axis of the tensor product: out[o][i][q] = sum_j w[j][i] * in[o][idx[j][i]][q].  IDX = unsigned when the
// element counts fit 32 bits (64-bit division is emulated on the GPU).

// The contiguous axis (inner == 1): neighbouring lanes own neighbouring output points i of one line, so the table
// entries (entry-major tables) and the outputs are coalesced and the inputs of a wave lie within a few cache
// lines.  All W table entries and inputs are loaded before the first is used (padded entries point at a valid
// input and are left out of the sum by a select - same bits as skipping them).  WT: compile-time width, 0 = any.
template <class IDX, int WT>
__global__ __launch_bounds__(256) void k_xfer_line(XferArgs a) {
    const IDX n_out = (IDX)a.n_out;
    const IDX total = (IDX)a.outer * n_out;
    const int W = WT ? WT : a.W;
    for (IDX p = blockIdx.x * (IDX)blockDim.x + threadIdx.x; p < total; p += (IDX)gridDim.x * blockDim.x) {
        const IDX o = p / n_out;
        const IDX i = p - o * n_out;
        const double* __restrict__ src = a.in + (size_t)o * a.n_in;
        double acc = 0.0;
        if (WT) {
            double wv[WT ? WT : 1], v[WT ? WT : 1];
#pragma unroll
            for (int j = 0; j < WT; ++j) {
                wv[j] = a.w[(size_t)j * n_out + i];
                v[j] = src[a.idx[(size_t)j * n_out + i]];
            }
#pragma unroll
            for (int j = 0; j < WT; ++j) acc = wv[j] != 0.0 ? acc + wv[j] * v[j] : acc;
        } else {
            for (int j = 0; j < W; ++j) {
                const double wj = a.w[(size_t)j * n_out + i];
                const double vj = src[a.idx[(size_t)j * n_out + i]];
                acc = wj != 0.0 ? acc + wj * vj : acc;
            }
        }
        a.out[p] = a.accumulate ? a.out[p] + acc : acc;
    }
}

// A strided axis, one thread per (o, q) walking CH consecutive output rows i: neighbouring rows read almost the
// same input rows, which then come from L1 instead of being fetched once per output row.  (Fallback for odd or
// unaligned inner sizes; the usual strided pass is k_xfer_axis_rows.)
template <class IDX, int CH>
__global__ __launch_bounds__(256) void k_xfer_axis(XferArgs a) {
    const IDX inner = (IDX)a.inner, outer = (IDX)a.outer;
    const IDX nch = ((IDX)a.n_out + CH - 1) / CH;
    const IDX total = outer * nch * inner;
    for (IDX p = blockIdx.x * (IDX)blockDim.x + threadIdx.x; p < total; p += (IDX)gridDim.x * blockDim.x) {
        const IDX r = p / inner;
        const IDX q = p - r * inner;
        const IDX o = r / nch;
        const int i0 = (int)(r - o * nch) * CH;
        const double* __restrict__ src = a.in + ((size_t)o * a.n_in) * a.inner + q;
        double* __restrict__ dst = a.out + ((size_t)o * a.n_out) * a.inner + q;
#pragma unroll 1
        for (int i = i0; i < i0 + CH && i < a.n_out; ++i) {
            double acc = 0.0;
            for (int j = 0; j < a.W; ++j) {
                const double wj = a.w[(size_t)j * a.n_out + i];
                if (wj != 0.0) acc += wj * src[(size_t)a.idx[(size_t)j * a.n_out + i] * a.inner];
            }
            dst[(size_t)i * a.inner] = a.accumulate ? dst[(size_t)i * a.inner] + acc : acc;
        }
    }
}

// The same pass for a strided axis (inner >= 2, even) with the output row taken from the block index: row index,
// table entries and weights are wave-uniform (scalar loads), every thread moves two neighbouring q (16-byte accesses)
// and CH consecutive output rows whose input rows overlap (refinement) are served by L1.
//   grid.x = nq * outer * ceil(n_out / CH) with nq blocks along inner / 2 (fastest)
template <int CH>
__global__ __launch_bounds__(256) void k_xfer_axis_rows(XferArgs a, unsigned nq) {
    const unsigned nch = (unsigned)(a.n_out + CH - 1) / CH;
    const unsigned by = blockIdx.x / nq, bx = blockIdx.x - by * nq;
    const unsigned o = by / nch, i0 = (by - o * nch) * CH;
    const size_t q2 = bx * (size_t)blockDim.x + threadIdx.x;  // pair index along the contiguous direction
    if (2 * q2 >= a.inner) return;
    const double2* __restrict__ src = reinterpret_cast<const double2*>(a.in + ((size_t)o * a.n_in) * a.inner) + q2;
    double2* __restrict__ dst = reinterpret_cast<double2*>(a.out + ((size_t)o * a.n_out) * a.inner) + q2;
    const size_t row = a.inner >> 1;
#pragma unroll
    for (int r = 0; r < CH; ++r) {
        const int i = (int)i0 + r;
        if (i < a.n_out) {
            double2 acc = double2{0.0, 0.0};
            for (int j = 0; j < a.W; ++j) {
                const double wj = a.w[(size_t)j * a.n_out + i];
                if (wj != 0.0) {
                    const double2 v = src[(size_t)a.idx[(size_t)j * a.n_out + i] * row];
                    acc.x += wj * v.x;
                    acc.y += wj * v.y;
                }
            }
            if (a.accumulate) {
                const double2 old = dst[(size_t)i * row];
                acc.x = old.x + acc.x;
                acc.y = old.y + acc.y;
            }
            dst[(size_t)i * row] = acc;
        }
    }
}

// Restriction in 3-D as ONE launch: out[o][i][j][k] = sum_a w[a][i] ( sum_b w[b][j] ( sum_c w[c][k] in[o][idx[a][i]][idx[b][j]][idx[c][k]] ) )
// with the sums nested and accumulated exactly like the three separable passes (contiguous axis innermost = first pass), so
// the result has their bits; a coarse point reads its (at most W^3 = 27 for rorder 2) fine neighbours through the caches
// instead of two intermediate fields being written and read: 256^3 -> 128^3, four fields: 343 -> ~120 us.
template <int W>
__global__ __launch_bounds__(256) void k_xfer_fused3(XferArgs a, unsigned nfields) {
    // grid: x = segments of the contiguous output axis, y = j, z = i + n_out * field: no integer divisions, and the table
    // entries of i and j are uniform over the workgroup (scalar loads)
    const unsigned n_out = (unsigned)a.n_out, n_in = (unsigned)a.n_in;
    const unsigned k = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
    const unsigned o = blockIdx.z / n_out, i = blockIdx.z - o * n_out;
    (void)nfields;
    if (k >= n_out) return;
    const double* __restrict__ src = a.in + (size_t)o * n_in * n_in * n_in;
    double wk[W];
    unsigned ik[W];
#pragma unroll
    for (int c = 0; c < W; ++c) {
        wk[c] = a.w[(size_t)c * n_out + k];
        ik[c] = (unsigned)a.idx[(size_t)c * n_out + k];
    }
    double acc0 = 0.0;
#pragma unroll
    for (int aa = 0; aa < W; ++aa) {
        const double wi = a.w[(size_t)aa * n_out + i];
        if (wi != 0.0) {
            const double* __restrict__ plane = src + (size_t)a.idx[(size_t)aa * n_out + i] * n_in * n_in;
            double acc1 = 0.0;
#pragma unroll
            for (int b = 0; b < W; ++b) {
                const double wj = a.w[(size_t)b * n_out + j];
                if (wj != 0.0) {
                    const double* __restrict__ row = plane + (size_t)a.idx[(size_t)b * n_out + j] * n_in;
                    double acc2 = 0.0;
#pragma unroll
                    for (int c = 0; c < W; ++c) {
                        const double v = row[ik[c]];
                        acc2 = wk[c] != 0.0 ? acc2 + wk[c] * v : acc2;
                    }
                    acc1 += wj * acc2;
                }
            }
            acc0 += wi * acc1;
        }
    }
    a.out[(((size_t)o * n_out + i) * n_out + j) * n_out + k] = acc0;
}

// ------------------------------------------------------------------------------------------------------
// Nested periodic grids, factor two per axis, 3-D, the table entries of every output row inside that row's window (checked
// by the host, sdc_transfer_apply_nested): one launch per transfer whose HBM traffic is the fields themselves.
// ------------------------------------------------------------------------------------------------------
struct NestedArgs {
    const double* in;
    const double* in_minus;    // refinement: in - in_minus is transferred (null: in)
    double* out;
    const double* out_minus;   // coarsening: result - out_minus is stored (null: result)
    const int* idx;
    const double* w;
    int n_out, n_in, W, accumulate;
};

// Coarsening with three entries per row (rorder 2: full weighting).  A thread owns the column (j, k) of the coarse grid and
// walks along i: the fine values 2k, 2k+1 of a row are ONE 16-byte load, 2k-1 comes from the neighbouring lane, so a wave
// reads whole 1 KB pieces of the fine rows; the plane sum of plane 2i+1 is kept for output i+1.  Per coarse value 2 planes x 3
// rows are loaded (the rows two neighbouring j share come from the caches); the sums are nested and ordered like the table
// says - what k_xfer_fused3 and the three separable passes compute.
//   grid: x = (k blocks) * (n_out / blockDim.y), y = n_out / TI, z = field;  block = (KX, JY), KX a power of two >= 16
#ifndef SDC_RESTRICT_THREADS
#define SDC_RESTRICT_THREADS 256
#endif
template <int TI>
__global__ __launch_bounds__(SDC_RESTRICT_THREADS) void k_restrict3_nested(NestedArgs a, unsigned kx_blocks) {
    const unsigned n_out = (unsigned)a.n_out, n_in = (unsigned)a.n_in;
#ifndef SDC_XFER_XCD
#define SDC_XFER_XCD 1
#endif
    // Workgroups are handed to the 8 XCDs round robin (x fastest): with the plain order the two workgroups that share a fine
    // row sit on different XCDs and each L2 fetches it.  Every XCD takes a contiguous eighth of the j blocks instead.
    unsigned bxs = blockIdx.x;
    if (SDC_XFER_XCD && (gridDim.x & 7u) == 0) bxs = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const unsigned kb = bxs % kx_blocks, jb = bxs / kx_blocks;
    const unsigned k = kb * blockDim.x + threadIdx.x, j = jb * blockDim.y + threadIdx.y;
    const unsigned i0 = blockIdx.y * TI, o = blockIdx.z;
    const double* __restrict__ src = a.in + (size_t)o * n_in * n_in * n_in;
    double wk[3], wj[3];
    int ck[3];        // which of (left, 2k, 2k+1) entry c reads
    unsigned rj[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        wk[c] = a.w[(size_t)c * n_out + k];
        const unsigned ik = (unsigned)a.idx[(size_t)c * n_out + k];
        ck[c] = ik == 2 * k ? 1 : (ik == 2 * k + 1 ? 2 : 0);
        wj[c] = a.w[(size_t)c * n_out + j];
        rj[c] = (unsigned)a.idx[(size_t)c * n_out + j];
    }
    const unsigned kl = (2 * k + n_in - 1) % n_in;
    // (the lane to the left holds column k - 1 of the same row unless this one starts a row or a wave)
    const bool lane0 = threadIdx.x == 0 || ((threadIdx.y * blockDim.x + threadIdx.x) & 63u) == 0;
    auto row_sum = [&](const double* __restrict__ row) {
#ifndef SDC_RESTRICT_NT
#define SDC_RESTRICT_NT 0
#endif
        const double2* __restrict__ pv = reinterpret_cast<const double2*>(row + 2 * (size_t)k);
        const double2 v = SDC_RESTRICT_NT ? double2{__builtin_nontemporal_load(&pv->x), __builtin_nontemporal_load(&pv->y)} : *pv;
        double left = __shfl_up(v.y, 1);
        if (lane0) left = row[kl];
        double acc = 0.0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double val = ck[c] == 1 ? v.x : (ck[c] == 2 ? v.y : left);
            acc = wk[c] != 0.0 ? acc + wk[c] * val : acc;
        }
        return acc;
    };
    auto plane_sum = [&](unsigned p) {
        const double* __restrict__ pl = src + (size_t)p * n_in * n_in;
        double s[3];
#pragma unroll
        for (int b = 0; b < 3; ++b) s[b] = row_sum(pl + (size_t)rj[b] * n_in);   // (all three rows in flight)
        double acc = 0.0;
#pragma unroll
        for (int b = 0; b < 3; ++b) acc = wj[b] != 0.0 ? acc + wj[b] * s[b] : acc;
        return acc;
    };
#ifndef SDC_RESTRICT_UNROLL
#define SDC_RESTRICT_UNROLL 1
#endif
    double tm = plane_sum((2 * i0 + n_in - 1) % n_in);
#pragma unroll SDC_RESTRICT_UNROLL
    for (unsigned ii = 0; ii < (unsigned)TI; ++ii) {
        const unsigned i = i0 + ii;
        const double t0 = plane_sum(2 * i), tp = plane_sum(2 * i + 1);
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const double wi = a.w[(size_t)q * n_out + i];
            const unsigned ip = (unsigned)a.idx[(size_t)q * n_out + i];
            const double tv = ip == 2 * i ? t0 : (ip == 2 * i + 1 ? tp : tm);
            acc = wi != 0.0 ? acc + wi * tv : acc;
        }
        const size_t q = (((size_t)o * n_out + i) * n_out + j) * n_out + k;
        if (a.out_minus) acc = acc - a.out_minus[q];
        a.out[q] = acc;
        tm = tp;
    }
}

// Refinement (iorder = W: even rows copy a coarse value, odd rows interpolate W of them).  A workgroup of 256 threads makes a
// 16^3 tile of the fine grid from the (8 + W - 1)^3 coarse values it depends on, staged in LDS.  The window is worked off
// plane by plane along the first axis: contiguous axis (T1: one plane of (8 + W - 1) x 16 values in LDS, two buffers), then
// the middle axis, whose result - one value per thread (jf, kf) and plane - stays in REGISTERS; the first axis then is a sum
// over a thread's own registers, stored or added to what is there (the coarse-grid correction).  The three separable passes
// with their intermediate fields on chip; 21 KB of LDS per workgroup, so that five or six of them share a CU and the
// read-modify-write of one tile runs behind the sums of the others.
// Even fine indices copy ONE coarse value, odd ones interpolate W: the threads are arranged so that a wave holds rows of one
// parity and skips the entries with weight zero by a branch that is uniform over the wave.  Sums along the contiguous and
// the middle axis run in table order; along the first axis in the order of the window (the table's order except where the
// window wraps around the periodic seam - the same terms).
//   grid: x = tiles along the contiguous axis, y = tiles along the middle axis, z = tiles along the first axis * fields
template <int W>
__global__ __launch_bounds__(256) void k_prolong3_nested(NestedArgs a) {
    constexpr int TC = 8, TF = 16, HW = TC + W - 1, HL = W / 2 - 1;
    __shared__ double A[HW * HW * HW];     // coarse window [x][y][z]
    __shared__ double T1[2][HW * TF];      // one x plane after the contiguous axis: [y][kf]
    __shared__ double WX[TF * W];          // rows of the first axis for the 16 fine planes of the tile, by window slot
    const int n_out = a.n_out, n_in = a.n_in;
    const int tiles = n_in / TC;
    int tz = blockIdx.x, ty = blockIdx.y;
    if (SDC_XFER_XCD && ((gridDim.x * gridDim.y) & 7u) == 0) {   // every XCD a contiguous eighth of the (ty, tz) tiles of a slab
        const unsigned lin = blockIdx.x + gridDim.x * blockIdx.y, per = (gridDim.x * gridDim.y) >> 3;
        const unsigned l2 = (lin & 7u) * per + (lin >> 3);
        tz = (int)(l2 % gridDim.x);
        ty = (int)(l2 / gridDim.x);
    }
    const int tx = blockIdx.z % tiles, o = blockIdx.z / tiles;
    const int tid = threadIdx.x, wv = tid >> 6, ln = tid & 63, par = wv & 1;
    const int kf1 = 2 * (ln & 7) + par, y1 = (ln >> 3) + 8 * (wv >> 1);          // first stage: row y1 (< HW) of the plane
    const int kf = ln & 15, jf = 2 * ((ln >> 4) + 4 * (wv >> 1)) + par;          // second / third stage
    const double* __restrict__ src = a.in + (size_t)o * n_in * n_in * n_in;
    const double* __restrict__ sub = a.in_minus ? a.in_minus + (size_t)o * n_in * n_in * n_in : nullptr;
    // window origin per axis (coarse index of local position 0); every sum below lies in [0, 2 n_in): one conditional
    // subtraction instead of a division
    auto wrap = [n_in](int v) { return v >= n_in ? v - n_in : v; };
    const int ox = wrap(tx * TC - HL + n_in), oy = wrap(ty * TC - HL + n_in), oz = wrap(tz * TC - HL + n_in);
    for (int e = tid; e < HW * HW * HW; e += 256) {
        const int z = e % HW, y = (e / HW) % HW, x = e / (HW * HW);
        const size_t g = ((size_t)wrap(ox + x) * n_in + (size_t)wrap(oy + y)) * n_in + (size_t)wrap(oz + z);
        double v = src[g];
        if (sub) v = v - sub[g];
        A[e] = v;
    }
    if (tid < TF * W) WX[tid] = 0.0;
    // this thread's table rows along the contiguous (kf1) and the middle (jf) axis: weights and LOCAL window positions
    double wz[W], wy[W];
    int lz[W], ly[W];
    {
        const int kg = tz * TF + kf1, jg = ty * TF + jf;
#pragma unroll
        for (int c = 0; c < W; ++c) {
            wz[c] = a.w[(size_t)c * n_out + kg];
            lz[c] = wrap(a.idx[(size_t)c * n_out + kg] - oz + n_in);
            wy[c] = a.w[(size_t)c * n_out + jg];
            ly[c] = wrap(a.idx[(size_t)c * n_out + jg] - oy + n_in) * TF;
        }
    }
    double* __restrict__ dst = a.out + (size_t)o * n_out * n_out * n_out +
                               ((size_t)(tx * TF) * n_out + (size_t)(ty * TF + jf)) * n_out + (size_t)(tz * TF + kf);
    const size_t plane = (size_t)n_out * n_out;
#ifndef SDC_PROLONG_OLD_LATE
#define SDC_PROLONG_OLD_LATE 0
#endif
    // (the fine values the tile is added to are fetched now: their latency hides behind everything below)
    double old[TF];
#ifndef SDC_PROLONG_NT
#define SDC_PROLONG_NT 3   // nontemporal load / store of the fine tile that is added to (bit 0 / 1): 272 -> 240 us at 3 x 128^3 -> 256^3
#endif
    if (!SDC_PROLONG_OLD_LATE && a.accumulate) {
#pragma unroll
        for (int fi = 0; fi < TF; ++fi) old[fi] = (SDC_PROLONG_NT & 1) ? __builtin_nontemporal_load(dst + fi * plane) : dst[fi * plane];
    }
    __syncthreads();
    if (tid < TF * W) {   // entry c of fine plane fi sits in window slot (local position) - fi / 2
        const int fi = tid / W, c = tid % W, ig = tx * TF + fi;
        const double w = a.w[(size_t)c * n_out + ig];
        if (w != 0.0) WX[fi * W + wrap(a.idx[(size_t)c * n_out + ig] - ox + n_in) - (fi >> 1)] = w;
    }
    double t2[HW];
#pragma unroll
    for (int x = 0; x < HW; ++x) {
        if (y1 < HW) {
            const double* __restrict__ row = A + (x * HW + y1) * HW;
            double acc = 0.0;
#pragma unroll
            for (int c = 0; c < W; ++c)
                if (wz[c] != 0.0) acc += wz[c] * row[lz[c]];
            T1[x & 1][y1 * TF + kf1] = acc;
        }
        __syncthreads();   // (two buffers: plane x + 1 is written while stragglers still read plane x - never plane x + 2)
        const double* __restrict__ pl = T1[x & 1] + kf;
        double acc = 0.0;
#pragma unroll
        for (int c = 0; c < W; ++c)
            if (wy[c] != 0.0) acc += wy[c] * pl[ly[c]];
        t2[x] = acc;
    }
    if (SDC_PROLONG_OLD_LATE && a.accumulate) {
#pragma unroll
        for (int fi = 0; fi < TF; ++fi) old[fi] = dst[fi * plane];
    }
#pragma unroll
    for (int fi = 0; fi < TF; ++fi) {
        double acc = 0.0;
#pragma unroll
        for (int p = 0; p < W; ++p) {
            // (the row of the first axis is the same for the whole workgroup: scalar weight, scalar branch)
            const double wl = WX[fi * W + p];
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)__double2loint(wl));
            const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)__double2hiint(wl));
            const double wi = __hiloint2double((int)hi, (int)lo);
            if (wi != 0.0) acc += wi * t2[(fi >> 1) + p];
        }
        const double res = a.accumulate ? old[fi] + acc : acc;
        if (SDC_PROLONG_NT & 2) __builtin_nontemporal_store(res, dst + fi * plane);
        else dst[fi * plane] = res;
    }
}

// ------------------------------------------------------------------------------------------------------
// Fourier prolongation (mesh_to_mesh_fft, TransferMesh_FFT.py:36-57; mesh_to_mesh_fft2d,
// TransferMesh_FFT2D.py:58-77): the coarse spectrum is copied into the low modes of a fine spectrum.  The
// reference's index conventions are kept as they are, including where the coarse Nyquist mode ends up.
// ------------------------------------------------------------------------------------------------------

// 1-D: fine_hat[0:h] = coarse_hat[0:h], fine_hat[nf/2] = coarse_hat[nc/2] (h = nc/2), irfft.  Both engines
// hold FULL complex spectra of the promoted line; irfft ignores the imaginary parts of DC and Nyquist.
__global__ void k_pad_spectrum_1d(const cd* __restrict__ G, cd* __restrict__ T, int nc, int nf) {
    const int h = nc / 2;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < nf; k += gridDim.x * blockDim.x) {
        const int kk = k <= nf / 2 ? k : nf - k;
        cd v = cd{0.0, 0.0};
        if (kk < h) v = G[kk];
        else if (kk == nf / 2) v = G[h];
        if (kk == 0 || kk == nf / 2) v.y = 0.0;
        if (k > nf / 2) v.y = -v.y;
        T[k] = v;
    }
}

// 2-D: the four corner blocks of fft2(G) go to the four corners of a zero fine spectrum, real(ifft2).  The
// fine engine transforms back from the half spectrum kx <= nf/2 of a real field, i.e. of the Hermitian part
// H(a,b) = (T(a,b) + conj T(-a,-b)) / 2 of the padded spectrum T - which is what taking the real part does.
// G: coarse half spectrum [nc/2+1][nc] (fft2 of a real field: G(-a,-b) = conj G(a,b)).
__device__ __forceinline__ cd pad2d_T(const cd* __restrict__ G, int nc, int nf, int a, int b) {
    const int h = nc / 2;
    const bool ina = a < h || a >= nf - h, inb = b < h || b >= nf - h;
    if (!ina || !inb) return cd{0.0, 0.0};
    const int ca = a < h ? a : a - (nf - nc), cb = b < h ? b : b - (nf - nc);
    if (ca <= nc / 2) return G[(size_t)ca * nc + cb];
    const cd g = G[(size_t)(nc - ca) * nc + ((nc - cb) % nc)];
    return cd{g.x, -g.y};
}
__global__ void k_pad_spectrum_2d(const cd* __restrict__ G, cd* __restrict__ T, int nc, int nf) {
    const int rows = nf / 2 + 1;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < rows * nf; p += gridDim.x * blockDim.x) {
        const int a = p / nf, b = p % nf;
        const cd t0 = pad2d_T(G, nc, nf, a, b);
        const cd t1 = pad2d_T(G, nc, nf, (nf - a) % nf, (nf - b) % nf);
        T[p] = cd{0.5 * (t0.x + t1.x), 0.5 * (t0.y - t1.y)};
    }
}

// 3-D: the eight corner blocks of fftn(G) go to the eight corners of a zero fine spectrum, real(ifftn) - the 2-D rule of
// TransferMesh_FFT2D.py:58-77 with one more axis (a field that does not depend on one axis is prolonged plane by plane
// exactly as mesh_to_mesh_fft2d does it: tests/test_gpu_multilevel.py).  G: coarse half spectrum [nc/2+1][nc][nc].
__device__ __forceinline__ cd pad3d_T(const cd* __restrict__ G, int nc, int nf, int a, int b, int c) {
    const int h = nc / 2;
    const bool ina = a < h || a >= nf - h, inb = b < h || b >= nf - h, inc = c < h || c >= nf - h;
    if (!ina || !inb || !inc) return cd{0.0, 0.0};
    const int ca = a < h ? a : a - (nf - nc), cb = b < h ? b : b - (nf - nc), cc = c < h ? c : c - (nf - nc);
    if (ca <= nc / 2) return G[((size_t)ca * nc + cb) * nc + cc];
    const cd g = G[((size_t)(nc - ca) * nc + ((nc - cb) % nc)) * nc + ((nc - cc) % nc)];
    return cd{g.x, -g.y};
}
__global__ void k_pad_spectrum_3d(const cd* __restrict__ G, cd* __restrict__ T, int nc, int nf) {
    const size_t total = (size_t)(nf / 2 + 1) * nf * nf;
    for (size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x; p < total; p += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(p % nf), b = (int)((p / nf) % nf), a = (int)(p / ((size_t)nf * nf));
        const cd t0 = pad3d_T(G, nc, nf, a, b, c);
        const cd t1 = pad3d_T(G, nc, nf, (nf - a) % nf, (nf - b) % nf, (nf - c) % nf);
        T[p] = cd{0.5 * (t0.x + t1.x), 0.5 * (t0.y - t1.y)};
    }
}
