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
