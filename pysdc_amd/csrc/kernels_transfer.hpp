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
    const int* idx;     // [n_out][W] source indices along the axis (device)
    const double* w;    // [n_out][W] weights (zero-padded)
    size_t outer, inner;
    int n_out, n_in, W;
};

// one axis of the tensor product: out[o][i][q] = sum_j w[i][j] * in[o][idx[i][j]][q]
__global__ void k_xfer_axis(XferArgs a) {
    const size_t total = a.outer * a.n_out * a.inner;
    for (size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x; p < total; p += (size_t)gridDim.x * blockDim.x) {
        const size_t q = p % a.inner;
        const size_t r = p / a.inner;
        const int i = (int)(r % a.n_out);
        const size_t o = r / a.n_out;
        const double* __restrict__ src = a.in + o * a.n_in * a.inner + q;
        double acc = 0.0;
        for (int j = 0; j < a.W; ++j) {
            const double wj = a.w[i * a.W + j];
            if (wj != 0.0) acc += wj * src[(size_t)a.idx[i * a.W + j] * a.inner];
        }
        a.out[p] = acc;
    }
}

