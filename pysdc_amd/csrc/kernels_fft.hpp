// libsdcmi kernels: axis passes of the 3-D real FFT, node-coupled Fourier solve, spectral-cache sweep.
#pragma once
#include "context.hpp"

// ------------------------------------------------------------------------------------------------------
// FFT kernels
// ------------------------------------------------------------------------------------------------------
struct FieldPtrs {
    const double* in[MAXM];
    double* out[MAXM];
};

// A pointwise reaction term evaluated where the values pass through registers anyway (Allen-Cahn: the explicit part of
// imex right-hand sides, AllenCahn_MPIFFT.py:83-85 / AllenCahn_2D_FFT.py): of the field the first pass READS (eval_f of a
// given u) or of the field the last pass WRITES (the node value a sweep just solved for).  out == null: nothing.
struct ReactEpi {
    double* out;
    int field, kind, nu;   // field < 0: every field f of the launch, into outs[f]
    double p0, p1;
    double* outs[MAXM];
    // The implicit part at the value a node solve just produced, without a transform of its own: (1 - alpha A) u = rhs gives
    // A u = (u - rhs) / alpha (generic_MPIFFT_Laplacian.py:164-211 evaluates it through a second transform round trip).  The
    // last pass of the solve has u in registers and reads rhs from `rhs` (which may be the field u is written to: every
    // thread reads its elements before it stores them).  impl_out == null: nothing.  Field 0 of the launch.
    double* impl_out;
    const double* rhs;
    double inv_alpha;
    __host__ __device__ double* target(int f) const { return field < 0 ? outs[f] : (f == field ? out : nullptr); }
};
__device__ __forceinline__ double react_value(double v, int kind, double p0, double p1, int nu) {
#pragma clang fp contract(off)
    if (kind == 1) {
        double pw = 1.0;
        for (int q = 0; q < nu; ++q) pw *= v;
        return p0 * v * (1.0 - pw);
    }
    return p0 * v * (1.0 - v) * (1.0 - 2.0 * v) - p1 * v * (1.0 - v);
}

// Terms added to the field a forward transform reads, on the way in: in[0] + sum_k c[k] x[k] - the right-hand side of one node
// of the reference's loop (generic_implicit.py:87-89 / imex_1st_order.py:92-94: gathered part + dt QI[m][j] f_j (+ dt QE[m][j]
// f_j.expl) of the nodes before it) without a pass of its own.  n = 0: nothing.
struct LinTerms {
    const double* x[2 * MAXM];
    double c[2 * MAXM];
    int n;
    double* wb;   // != null: the completed field is stored there as well (the right-hand side the last pass of the solve reads again)
};

// 1-D problems: promote the real line to complex / take the real part back
__global__ void k_promote(FieldPtrs p, cd* W, size_t N) {
    const int f = blockIdx.y;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < N; i += (size_t)gridDim.x * blockDim.x)
        W[(size_t)f * N + i] = cd{p.in[f][i], 0.0};
}
__global__ void k_realpart(FieldPtrs p, const cd* W, size_t N) {
    const int f = blockIdx.y;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < N; i += (size_t)gridDim.x * blockDim.x)
        p.out[f][i] = W[(size_t)f * N + i].x;
}

// r2c along axis 0: real field [N][rest] seen as complex pairs [N][rest/2]; two real columns per complex
// column ("two for one"), unpacked to the half spectra W[k][rest], k = 0..N/2.
template <int N, int T>
__global__ __launch_bounds__((N / fft_elems(N)) * T, 4) void k_fftx_fwd(FieldPtrs p, cd* __restrict__ W, size_t fstride,
                                                                      int rest, const cd* __restrict__ tw,
                                                                      ReactEpi epi = ReactEpi{nullptr, 0, 0, 0, 0.0, 0.0, {}},
                                                                      LinTerms lin = LinTerms{{}, {}, 0}) {
    constexpr int E = fft_elems(N), P = N / E;
    using LAY = LayStrided<N, T>;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int col = threadIdx.x % T, j = threadIdx.x / T;
    const int ncol = rest >> 1;  // complex columns
    const int c = blockIdx.x * T + col;
    const bool ok = c < ncol;
    const double* __restrict__ in = p.in[blockIdx.y];
#ifndef SDC_XFWD_NT
#define SDC_XFWD_NT 1   // bit 0: nontemporal loads of the fields read (round 6, 256^3: the node right-hand sides with their terms 153 -> 130 us;
                        // neutral at 1024^3), bit 1: nontemporal stores of the spectrum (a loss: the next pass reads it)
#endif
    auto ldf = [](const double* q) {
#if SDC_XFWD_NT & 1
        return cd{__builtin_nontemporal_load(q), __builtin_nontemporal_load(q + 1)};
#else
        return *reinterpret_cast<const cd*>(q);
#endif
    };
    cd r[E];
#pragma unroll
    for (int i = 0; i < E; ++i)
        r[i] = ok ? ldf(in + (size_t)(j + i * P) * rest + 2 * (size_t)c) : cd{0.0, 0.0};
    if (lin.n > 0 && blockIdx.y == 0 && ok) {   // (same accumulation order as k_lincomb: base, then term by term)
        for (int k = 0; k < lin.n; ++k) {
            const double* __restrict__ xk = lin.x[k];
            const double ck = lin.c[k];
#pragma unroll
            for (int i = 0; i < E; ++i) {
                const cd v = ldf(xk + (size_t)(j + i * P) * rest + 2 * (size_t)c);
                r[i].x += ck * v.x;
                r[i].y += ck * v.y;
            }
        }
        if (lin.wb) {
#pragma unroll
            for (int i = 0; i < E; ++i) *reinterpret_cast<cd*>(lin.wb + (size_t)(j + i * P) * rest + 2 * (size_t)c) = r[i];
        }
    }
    if (double* const eo = epi.target((int)blockIdx.y); eo && ok) {  // reaction term of the field that is being read
#pragma unroll
        for (int i = 0; i < E; ++i)
            *reinterpret_cast<cd*>(eo + (size_t)(j + i * P) * rest + 2 * (size_t)c) =
                cd{react_value(r[i].x, epi.kind, epi.p0, epi.p1, epi.nu), react_value(r[i].y, epi.kind, epi.p0, epi.p1, epi.nu)};
    }
    fft_line<N, -1, LAY>(r, j, col, lds, tw);
    // unpack: A[k] = (C[k] + conj C[N-k]) / 2, B[k] = (C[k] - conj C[N-k]) / (2i)
    cd A[E], B[E];
#pragma unroll
    for (int part = 0; part < 2; ++part) {
#pragma unroll
        for (int i = 0; i < E; ++i) lds[LAY::idx(col, j + i * P)] = part == 0 ? r[i].x : r[i].y;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < E; ++i) {
            const int k = j + i * P;
            const double v = lds[LAY::idx(col, (N - k) % N)];
            if (part == 0) {
                A[i].x = 0.5 * (r[i].x + v);
                B[i].y = -0.5 * (r[i].x - v);
            } else {
                A[i].y = 0.5 * (r[i].y - v);
                B[i].x = 0.5 * (r[i].y + v);
            }
        }
        __syncthreads();
    }
    cd* __restrict__ Wf = W + blockIdx.y * fstride;
#pragma unroll
    for (int i = 0; i < E; ++i) {
        const int k = j + i * P;
        if (ok && k <= N / 2) {
            cd* dst = Wf + (size_t)k * rest + 2 * (size_t)c;
#if SDC_XFWD_NT & 2
            __builtin_nontemporal_store(A[i].x, &dst[0].x);
            __builtin_nontemporal_store(A[i].y, &dst[0].y);
            __builtin_nontemporal_store(B[i].x, &dst[1].x);
            __builtin_nontemporal_store(B[i].y, &dst[1].y);
#else
            dst[0] = A[i];
            dst[1] = B[i];
#endif
        }
    }
}

// SDC_XINV_DIRECT=1: every thread fetches the mirrored rows itself instead of unpacking through LDS - measured slower
// at 1024^3 (12.4 vs 10.2 ms for the norm pass), kept as a build-time variant
#ifndef SDC_XINV_DIRECT
#define SDC_XINV_DIRECT 0
#endif
#ifndef SDC_XWAVE
#define SDC_XWAVE 1  // the norm-only pass transforms every column inside one wavefront (4 workgroup barriers per tile, not 12:
                     // 9.9 -> 9.7 ms at 1024^3); 0: the round-2 arrangement (columns spread over all waves)
#endif
// c2r along axis 0 (inverse of the above, unnormalised).  NORM: max |.| per field goes to norms[field] (the fields
// are the collocation residuals of the spectral sweep); STORE: the real field is written to out[field].
// ADD: a further half spectrum `add` (same layout, one field) is added to every field on the way in - the residual of
// all nodes changes by the same field when u[0] is replaced (time-parallel runs), and the transform is linear.
// SCR: the transformed tile of ONE field parked in / taken from a scratch field in the arrangement the threads hold it
// (scr[(tile * E + i) * threads + thread]: every access a contiguous 16 bytes per lane).  1: the tile is stored there and
// nothing else happens (the difference of two start values, time-parallel runs: the residual of EVERY node changes by that
// field when u[0] is replaced, core/sweeper.py:186-199 is linear in u[0]).  2: max |r| goes to norms2[field] and max |r + d|
// with d from the scratch to norms[field] - the residual before and after the receive out of one pass over the residual
// lines (same XCD-aware order as ADD: the nfields workgroups that read one scratch tile follow each other on one XCD).
// IMPL: the implicit part from the solve's own equation rides along (ReactEpi::impl_out; an instantiation of its own, so
// that the registers it needs are not everybody's).
template <int N, int T, bool NORM, bool STORE, bool ADD = false, int SCR = 0, bool IMPL = false>
__global__ __launch_bounds__((N / fft_elems(N)) * T, 4) void k_fftx_inv(FieldPtrs p, const cd* __restrict__ W,
                                                                      size_t fstride, int rest,
                                                                      const cd* __restrict__ tw,
                                                                      unsigned long long* __restrict__ norms,
                                                                      const cd* __restrict__ add = nullptr, int nfields = 1,
                                                                      ReactEpi epi = ReactEpi{nullptr, 0, 0, 0, 0.0, 0.0, {}},
                                                                      cd* __restrict__ scr = nullptr,
                                                                      unsigned long long* __restrict__ norms2 = nullptr) {
    constexpr int E = fft_elems(N), P = N / E;
    static_assert(SCR == 0 || (!STORE && !ADD && (SCR == 1) == !NORM), "scratch variants: 1 = store only, 2 = both norm sets");
    constexpr bool GRID1D = ADD || SCR == 2;
    constexpr bool XWAVE = SDC_XWAVE && ((NORM && !STORE) || SCR != 0) && P == 64 && !SDC_XINV_DIRECT;  // (nothing stored in real space: one wave per column)
    using LAY = LayStrided<N, T>;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int col = threadIdx.x % T, j = threadIdx.x / T;
    const int ncol = rest >> 1;
    // ADD: one-dimensional grid in which the NF workgroups that read the SAME tile of `add` follow each other at a
    // distance of 8 - workgroups are handed to the 8 XCDs round robin, so those NF land on one XCD and find the tile in
    // its L2 after the first of them fetched it (id = 8 NF g + 8 f + t: tile 8 g + t, field f)
#ifndef SDC_X_SWZ
#define SDC_X_SWZ 0  // 1: every XCD a contiguous eighth of the tiles (9.5 -> 9.6 - 9.8 ms); 3: every XCD whole rows of the (y, z)
                     // plane (9.56 - 9.60 -> 9.65 - 9.86 ms): measured in round 3, the plain order stays
#endif
#ifndef SDC_X_SWZ_STORE
#define SDC_X_SWZ_STORE 3  // ... of the launches that WRITE the real fields (eager node fields, end values, multi-level paths): a pass
                           // that reads and writes gains from XCDs that work apart (17.2 - 17.5 -> 16.8 ms for five 1024^3 fields)
#endif
    constexpr int XS = STORE ? SDC_X_SWZ_STORE : SDC_X_SWZ;
    int bx_ = (int)blockIdx.x;
    if (XS == 1 && (gridDim.x & 7u) == 0) bx_ = (int)((blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3));
    if (XS == 3 && !GRID1D) {  // every XCD whole rows of the (y, z) plane: XCD x takes the rows y = 8 g + x and walks along z
        constexpr unsigned ZT = (N / 2 + T - 1) / T;
        if (gridDim.x == (unsigned)N * ZT && (N & 7) == 0) {
            const unsigned xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
            bx_ = (int)(((slot / ZT) * 8u + xcd) * ZT + slot % ZT);
        }
    }
    const int bx = GRID1D ? (int)((blockIdx.x / (8 * nfields)) * 8 + blockIdx.x % 8) : bx_;
    const int by = GRID1D ? (int)((blockIdx.x / 8) % nfields) : (int)blockIdx.y;
    const int c = bx * T + col;
    const bool ok = c < ncol;
    if (SCR == 2 && bx * T >= ncol) return;   // (a padding workgroup of the last group of 8: no scratch tile behind it)
    const cd* __restrict__ Wf = W + by * fstride;
    cd r[E];
#if SDC_XINV_DIRECT
    // C[k] = A[k] + i B[k] (k <= N/2), C[N-k] = conj A[k] + i conj B[k]: every thread fetches the row it needs for each
    // of its elements itself (rows 1 .. N/2-1 are read twice per workgroup, the second time from cache) - no trip
    // through LDS before the transform
#pragma unroll
    for (int i = 0; i < E; ++i) {
        const int k = j + i * P;
        const bool up = k > N / 2;
        const int kk = up ? N - k : k;
        cd a = cd{0.0, 0.0}, b = cd{0.0, 0.0};
        if (ok) {
            const cd* src = Wf + (size_t)kk * rest + 2 * (size_t)c;
            a = src[0];
            b = src[1];
        }
        const bool edge = (k == 0) || (k == N / 2);
        if (edge) r[i] = cd{a.x, b.x};
        else if (!up) r[i] = cd{a.x - b.y, a.y + b.x};
        else r[i] = cd{a.x + b.y, -a.y + b.x};
    }
    fft_line<N, +1, LAY>(r, j, col, lds, tw);
#else
    cd A[E], B[E];
#pragma unroll
    for (int i = 0; i < E; ++i) {
        const int k = j + i * P;
        if (ok && k <= N / 2) {
            const cd* src = Wf + (size_t)k * rest + 2 * (size_t)c;
#ifndef SDC_XINV_LD_NT
#define SDC_XINV_LD_NT 1   // nontemporal loads of the spectra in the launches that STORE real fields, lines up to 256 modes
                           // (config 5: -1 % per step; at 1024^3 the store pass LOSES 13 %: the spectra are long gone from the caches there)
#endif
            if constexpr (STORE && SDC_XINV_LD_NT && N <= 256) {
                A[i] = cd{__builtin_nontemporal_load(&src[0].x), __builtin_nontemporal_load(&src[0].y)};
                B[i] = cd{__builtin_nontemporal_load(&src[1].x), __builtin_nontemporal_load(&src[1].y)};
            } else {
                A[i] = src[0];
                B[i] = src[1];
            }
            if constexpr (ADD) {
                const cd* s2 = add + (size_t)k * rest + 2 * (size_t)c;
                A[i] = cadd(A[i], s2[0]);
                B[i] = cadd(B[i], s2[1]);
            }
        } else {
            A[i] = B[i] = cd{0.0, 0.0};
        }
    }
    // C[k] = A[k] + i B[k] (k <= N/2), C[N-k] = conj A[k] + i conj B[k]
    if constexpr (XWAVE) {
        // One wavefront per column for the transform: the loads above are coalesced over (rows, columns) - 8 rows x 8
        // columns per wave - but a column whose 64 threads sit in ONE wave exchanges through LDS without workgroup
        // barriers.  So the unpacked line goes to LDS column by column (both halves: the thread that holds row k makes
        // C[k] and C[N-k]) and every wave picks up its column: 4 workgroup barriers per tile instead of 12.
        using LC = LayCols<N>;
        const int cw = threadIdx.x >> 6, lane = threadIdx.x & 63;
        // rows k = j + i P and N - k, P a multiple of 16: the skewed positions are affine in i (fft.hpp) - one index each,
        // compile-time offsets for the rest
        static_assert(P % 16 == 0, "XWAVE: 64 threads per column");
        constexpr int STEP = P + P / 16;
        const int i_own = LC::idx(col, j), i_mir = LC::idx(col, N - j), i_rd = LC::idx(cw, lane);
#pragma unroll
        for (int part = 0; part < 2; ++part) {
#pragma unroll
            for (int i = 0; i < E; ++i) {
                const int k = j + i * P;
                if (k <= N / 2) {
                    const bool edge = (k == 0) || (k == N / 2);
                    double own, mir;
                    if (part == 0) {
                        own = edge ? A[i].x : A[i].x - B[i].y;
                        mir = A[i].x + B[i].y;
                    } else {
                        own = edge ? B[i].x : A[i].y + B[i].x;
                        mir = -A[i].y + B[i].x;
                    }
                    lds[i_own + i * STEP] = own;
                    if (!edge) lds[i_mir - i * STEP] = mir;
                }
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < E; ++i) {
                const double v = lds[i_rd + i * STEP];
                if (part == 0) r[i].x = v;
                else r[i].y = v;
            }
            __syncthreads();
        }
        fft_line<N, +1, LC, true>(r, lane, cw, lds, tw);
    } else {
#pragma unroll
    for (int part = 0; part < 2; ++part) {
#pragma unroll
        for (int i = 0; i < E; ++i) {
            const int k = j + i * P;
            if (k <= N / 2) {
                const bool edge = (k == 0) || (k == N / 2);
                double own, mir;
                if (part == 0) {
                    own = edge ? A[i].x : A[i].x - B[i].y;
                    mir = A[i].x + B[i].y;
                } else {
                    own = edge ? B[i].x : A[i].y + B[i].x;
                    mir = -A[i].y + B[i].x;
                }
                // the thread's own rows stay in registers; only the mirrored half crosses threads (through LDS)
                if (part == 0) r[i].x = own;
                else r[i].y = own;
                if (!edge) lds[LAY::idx(col, N - k)] = mir;
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < E; ++i) {
            const int k = j + i * P;
            if (k > N / 2) {
                const double v = lds[LAY::idx(col, k)];
                if (part == 0) r[i].x = v;
                else r[i].y = v;
            }
        }
        __syncthreads();
    }
    fft_line<N, +1, LAY>(r, j, col, lds, tw);
    }
#endif
    if constexpr (SCR == 1) {
        cd* __restrict__ dst = scr + (size_t)bx * E * blockDim.x + threadIdx.x;
#pragma unroll
        for (int i = 0; i < E; ++i) dst[(size_t)i * blockDim.x] = r[i];
    }
    if constexpr (SCR == 2) {
        // the norms before the receive from r, the norms after it from r + d (d: the transformed difference of the two start
        // values, parked by the SCR == 1 launch in this very arrangement)
        const cd* __restrict__ src = scr + (size_t)bx * E * blockDim.x + threadIdx.x;
        double m0 = 0.0, m1 = 0.0;
        cd d0 = src[0];
#pragma unroll
        for (int i = 0; i < E; ++i) {
            const cd d = i == 0 ? d0 : src[(size_t)i * blockDim.x];
            m0 = fmax(m0, fmax(fabs(r[i].x), fabs(r[i].y)));
            m1 = fmax(m1, fmax(fabs(r[i].x + d.x), fabs(r[i].y + d.y)));
        }
        // (NaN: see below - one element of every thread tells; the difference field gets the same test)
        if (r[0].x != r[0].x || r[0].y != r[0].y) m0 = m1 = r[0].x + r[0].y;
        if (d0.x != d0.x || d0.y != d0.y) m1 = d0.x + d0.y;
        m0 = wave_max(m0);
        m1 = wave_max(m1);
        if ((threadIdx.x & 63) == 0) {
            atomic_max_abs(norms2 + by, m0);
            atomic_max_abs(norms + by, m1);
        }
    } else if constexpr (NORM) {
        double m = 0.0;  // columns beyond the edge were transformed from zeros
#pragma unroll
        for (int i = 0; i < E; ++i) m = fmax(m, fmax(fabs(r[i].x), fabs(r[i].y)));
        // fmax drops NaNs, np.max does not: every output of the transform is a sum over ALL inputs of its column pair (the
        // real parts of output 0 over the real parts only, its imaginary part over the imaginary ones), so a NaN anywhere
        // in the pair shows in one output of every thread - looking at one element is an exact test
        if (r[0].x != r[0].x || r[0].y != r[0].y) m = r[0].x + r[0].y;
        // (one guarded atomic per wave: reducing over the workgroup first was measured - 9.82 -> 10.17 ms at 1024^3, the
        // barrier keeps finished waves from retiring; the relaxed read in atomic_max_abs is not a hot spot)
        m = wave_max(m);
        if ((threadIdx.x & 63) == 0) atomic_max_abs(norms + by, m);
    }
    if constexpr (STORE) {
        double* out = p.out[by];   // (no __restrict__: epi.rhs may be this very field)
        if (ok) {
            if constexpr (IMPL) {
                if (epi.impl_out && by == 0) {   // A u = (u - rhs) / alpha, four elements at a time (few registers beside r[])
                    constexpr int CHK = E % 4 == 0 ? 4 : 1;
#pragma unroll
                    for (int i0 = 0; i0 < E; i0 += CHK) {
                        cd q[CHK];
#pragma unroll
                        for (int i = 0; i < CHK; ++i)
                            q[i] = *reinterpret_cast<const cd*>(epi.rhs + (size_t)(j + (i0 + i) * P) * rest + 2 * (size_t)c);
#pragma unroll
                        for (int i = 0; i < CHK; ++i) {
                            __builtin_nontemporal_store((r[i0 + i].x - q[i].x) * epi.inv_alpha,
                                                        epi.impl_out + (size_t)(j + (i0 + i) * P) * rest + 2 * (size_t)c);
                            __builtin_nontemporal_store((r[i0 + i].y - q[i].y) * epi.inv_alpha,
                                                        epi.impl_out + (size_t)(j + (i0 + i) * P) * rest + 2 * (size_t)c + 1);
                        }
                    }
                }
            }
#ifndef SDC_XINV_NT
#define SDC_XINV_NT 0   // nontemporal stores of the real fields
#endif
            auto stf = [](double* q, cd v) {
#if SDC_XINV_NT
                __builtin_nontemporal_store(v.x, q);
                __builtin_nontemporal_store(v.y, q + 1);
#else
                *reinterpret_cast<cd*>(q) = v;
#endif
            };
#pragma unroll
            for (int i = 0; i < E; ++i) stf(out + (size_t)(j + i * P) * rest + 2 * (size_t)c, r[i]);
            if (double* const eo = epi.target(by)) {  // reaction term of the field that is being written
#pragma unroll
                for (int i = 0; i < E; ++i)
                    stf(eo + (size_t)(j + i * P) * rest + 2 * (size_t)c,
                        cd{react_value(r[i].x, epi.kind, epi.p0, epi.p1, epi.nu), react_value(r[i].y, epi.kind, epi.p0, epi.p1, epi.nu)});
            }
        }
    }
}

// cos / sin (pi i / 16), i = 0..15
__device__ static const double kCosPiOver[16] = {1.0, 0.98078528040323044913, 0.92387953251128675613, 0.83146961230254523708,
                                                 0.70710678118654752440, 0.55557023301960222474, 0.38268343236508977173,
                                                 0.19509032201612826785, 0.0, -0.19509032201612826785, -0.38268343236508977173,
                                                 -0.55557023301960222474, -0.70710678118654752440, -0.83146961230254523708,
                                                 -0.92387953251128675613, -0.98078528040323044913};
__device__ static const double kSinPiOver[16] = {0.0, 0.19509032201612826785, 0.38268343236508977173, 0.55557023301960222474,
                                                 0.70710678118654752440, 0.83146961230254523708, 0.92387953251128675613,
                                                 0.98078528040323044913, 1.0, 0.98078528040323044913, 0.92387953251128675613,
                                                 0.83146961230254523708, 0.70710678118654752440, 0.55557023301960222474,
                                                 0.38268343236508977173, 0.19509032201612826785};
#ifndef SDC_XHALF
#define SDC_XHALF 0  // 1: norm-only c2r pass by the half-length transform (below) for N = 256, 512, 1024 - measured SLOWER
                     // at 1024^3 (12.9 ms against 10.2 ms of the two-for-one kernel), kept as an experiment switch
#endif
#ifndef SDC_XHALF_MIRROR
#define SDC_XHALF_MIRROR 1
#endif
#ifndef SDC_XHALF_T
#define SDC_XHALF_T 8
#endif
#ifndef SDC_XHALF_WAVES
#define SDC_XHALF_WAVES 4
#endif
// c2r along axis 0, norm only, by the half-length transform: with H = N/2 and the Hermitian rows X[0..H] of a column,
//   Z[k] = (X[k] + conj X[H-k]) + i e^{+2 pi i k/N} (X[k] - conj X[H-k]),  k = 0..H-1,
// the H-point inverse transform z = IFFT_H(Z) holds the real line as z[m] = x[2m] + i x[2m+1].  One column is ONE
// complex line of length H instead of half of a packed line of length N, so every input byte costs half the LDS
// traffic of the two-for-one kernel above (k_fftx_inv: ~10 bytes through LDS per byte from HBM, which is what bounds
// it at 1024^3) and the mirrored half needs no trip through LDS either: a thread fetches row H-k itself (the second
// request for a row comes from a thread of the same workgroup and is served from cache).  tw: e^{-2 pi i k/N}, k < N;
// twh: the table of the H-point transform.
template <int N, int T>
__global__ __launch_bounds__((N / 2 / fft_elems(N / 2)) * T, SDC_XHALF_WAVES) void k_fftx_norm_half(
    const cd* __restrict__ W, size_t fstride, int rest, const cd* __restrict__ tw, const cd* __restrict__ twh,
    unsigned long long* __restrict__ norms) {
    constexpr int H = N / 2, E = fft_elems(H), P = H / E;
    using LAY = LayStrided<H, T>;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int col = threadIdx.x % T, j = threadIdx.x / T;
    const int c = blockIdx.x * T + col;
    const bool ok = c < rest;
    const cd* __restrict__ Wf = W + blockIdx.y * fstride + c;
    // pre-twiddle e^{+2 pi i k/N}, k = j + i P: one table look-up (k = j) times the constant e^{+i pi i/E}
    const cd wj = ok ? tw[j] : cd{1.0, 0.0};
    auto pre = [&](int i, cd x, cd y) {
        const int k = j + i * P;
        if (k == 0) x.y = y.y = 0.0;  // rows 0 and H of a real line are real (c2r convention: imaginary parts dropped)
        const cd e = cd{x.x + y.x, x.y - y.y};  // X[k] + conj X[H-k]
        const cd d = cd{x.x - y.x, x.y + y.y};  // X[k] - conj X[H-k]
        const double ci = kCosPiOver[(i * 16) / E], si = kSinPiOver[(i * 16) / E];
        const cd w = cd{wj.x * ci + wj.y * si, wj.x * si - wj.y * ci};  // conj(tw[j]) * e^{+i pi i/E} = (c, s): real, imag
        const cd o = cd{d.x * w.x - d.y * w.y, d.x * w.y + d.y * w.x};
        return cd{e.x - o.y, e.y + o.x};
    };
    cd r[E];
#if SDC_XHALF_MIRROR
    // own rows from memory, the mirrored ones from the threads that loaded them (through LDS, one plane at a time);
    // row H, the partner of row 0, is fetched by the thread that owns row 0
#pragma unroll
    for (int i = 0; i < E; ++i) r[i] = ok ? Wf[(size_t)(j + i * P) * rest] : cd{0.0, 0.0};
    cd top = cd{0.0, 0.0};
    if (j == 0 && ok) top = Wf[(size_t)H * rest];
    double bx[E];
#pragma unroll
    for (int i = 0; i < E; ++i) lds[LAY::idx(col, j + i * P)] = r[i].x;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < E; ++i) {
        const int k = j + i * P;
        bx[i] = k == 0 ? top.x : lds[LAY::idx(col, H - k)];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < E; ++i) lds[LAY::idx(col, j + i * P)] = r[i].y;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < E; ++i) {
        const int k = j + i * P;
        const double by = k == 0 ? top.y : lds[LAY::idx(col, H - k)];
        r[i] = pre(i, r[i], cd{bx[i], by});
    }
    __syncthreads();
#else
#pragma unroll
    for (int i = 0; i < E; ++i) {
        const int k = j + i * P;
        const cd x = ok ? Wf[(size_t)k * rest] : cd{0.0, 0.0};
        const cd y = ok ? Wf[(size_t)(H - k) * rest] : cd{0.0, 0.0};
        r[i] = pre(i, x, y);
    }
#endif
    fft_line<H, +1, LAY>(r, j, col, lds, twh);
    double m = 0.0;  // columns beyond the edge were transformed from zeros
#pragma unroll
    for (int i = 0; i < E; ++i) {
        const double v0 = fabs(r[i].x), v1 = fabs(r[i].y);
        const double v = (v0 > v1 || v0 != v0) ? v0 : v1;
        m = (m > v || m != m) ? m : v;
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) atomic_max_abs(norms + blockIdx.y, m);
}

// c2c in place along the middle axis of W[f][kx][y][z] (3-D only): tile = all y x T z-columns
template <int N, int T, int DIR>
__global__ __launch_bounds__((N / fft_elems(N)) * T, 4) void k_ffty(cd* __restrict__ W, size_t fstride,
                                                                  const cd* __restrict__ tw, int kx0 = 0) {
    constexpr int E = fft_elems(N), P = N / E;
    using LAY = LayStrided<N, T>;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int col = threadIdx.x % T, j = threadIdx.x / T;
    // SDC_Y_SWZ: which tile a workgroup takes.  Workgroups are handed to the 8 XCDs round robin in the order x fastest, then y.
    // 0 (rounds 1 - 2): tile = blockIdx.x, plane = blockIdx.y - the 8 XCDs work on 8 adjacent 128-byte segments of the same
    // rows at the same time, i.e. all of them on the same few DRAM pages / channels.  3 (default since round 3): every XCD
    // streams through its own contiguous range of kx planes (the planes left over when their number is not a multiple of 8
    // keep the plain order) - `fft_y_inv[5]` at 1024^3 16.2 -> 15.1 ms on one box, 13.8 - 14.6 on another; 2: whole planes,
    // handed out in groups of 8 (15.3); 1: a contiguous eighth of the tiles of each plane (17.3 - 18.0: worse than plain).
    // The same idea was measured for the x pass (every XCD a contiguous eighth of the tiles: 9.5 -> 9.6 - 9.8 ms, SDC_X_SWZ)
    // and for the contiguous-axis launch (SDC_Z_SWZ: no difference) and left off there.
#ifndef SDC_Y_SWZ
#define SDC_Y_SWZ 3
#endif
    unsigned bx = blockIdx.x, by = blockIdx.y;
#if SDC_Y_SWZ == 1
    if ((gridDim.x & 7u) == 0) bx = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
#elif SDC_Y_SWZ == 2
    {
        const unsigned lin = blockIdx.x + gridDim.x * blockIdx.y, full = (gridDim.y >> 3) * 8u * gridDim.x;
        if (lin < full) {
            const unsigned xcd = lin & 7u, slot = lin >> 3;
            by = (slot / gridDim.x) * 8u + xcd;
            bx = slot % gridDim.x;
        }
    }
#elif SDC_Y_SWZ == 3
    {   // every XCD a contiguous range of planes
        const unsigned lin = blockIdx.x + gridDim.x * blockIdx.y, per = gridDim.y >> 3, full = per * 8u * gridDim.x;
        if (lin < full) {
            const unsigned xcd = lin & 7u, slot = lin >> 3;
            by = xcd * per + slot / gridDim.x;
            bx = slot % gridDim.x;
        }
    }
#endif
    const int c = bx * T + col;
    const bool ok = c < N;
    cd* __restrict__ base = W + blockIdx.z * fstride + (size_t)(by + kx0) * N * N + c;  // kx0: a launch per group of kx planes
    cd r[E];
#pragma unroll
    for (int i = 0; i < E; ++i) r[i] = ok ? ld_stream(base + (size_t)(j + i * P) * N) : cd{0.0, 0.0};
    fft_line<N, DIR, LAY>(r, j, col, lds, tw);
    if (ok) {
#pragma unroll
        for (int i = 0; i < E; ++i) st_stream(base + (size_t)(j + i * P) * N, r[i]);
    }
}

struct ZArgs {
    cd* W;
    size_t fstride;
    const cd *tw, *lamI, *lamE;  // lamE may be null
    double cI[MAXM][MAXM];       // strictly lower: dt*QI[m+1][j+1], j < m
    double cE[MAXM][MAXM];       // strictly lower: dt*QE[m+1][j+1]
    double alpha[MAXM];          // dt*QI[m+1][m+1]
    double invN;
    int nf, ndim, coupled;
    int apply;  // 1: multiply by the symbol (operator application) instead of dividing by 1 - alpha*symbol
    int dup;    // 1: several fields out of ONE input (field 0).  Solve (apply = 0): the solution, the implicit operator applied
                // to it and - nf = 3 - the explicit operator (lamE) applied to it; apply = 1, nf = 2: the implicit and the
                // explicit operator applied to the input (eval_f of a problem whose two parts are given by their symbols)
};

// forward FFT along the contiguous axis, node-coupled implicit solve in Fourier space, inverse FFT.
// One workgroup = LPB lines x all nf fields; column c = f*LPB + l occupies threads [c*P, (c+1)*P).
// After the forward transform the spectra go through LDS once more so that one thread holds ALL nf node
// values of a mode: the lower-triangular node coupling is then a register recurrence with wave-uniform
// coefficient indices (scalar kernarg loads, no per-lane table look-ups).
template <int N>
constexpr int z_lines_per_block() {
    constexpr int P = N / fft_elems(N);
    return P >= 64 ? 1 : 64 / P;
}

// elements per thread of the node-solve launch: lines of 64 .. 256 modes take 8 instead of 16 - twice the waves, two thirds of
// the registers: with 16 a 128^3 level launches ONE wave per SIMD (26 -> ?? us) and a 256^3 level holds three per SIMD
#ifndef SDC_ZSOLVE_E
#define SDC_ZSOLVE_E 8
#endif
#ifndef SDC_ZSOLVE_E_SMALL
#define SDC_ZSOLVE_E_SMALL SDC_ZSOLVE_E   // ... of lines of 64 / 128 modes
#endif
template <int N>
constexpr int zsolve_elems() {
    constexpr int want = N <= 128 ? SDC_ZSOLVE_E_SMALL : SDC_ZSOLVE_E;
    return ((N & (N - 1)) == 0 && N >= 64 && N <= 256 && want < fft_elems(N)) ? want : fft_elems(N);
}
template <int N>
constexpr int zsolve_lines() {
    constexpr int P = N / zsolve_elems<N>();
    return P >= 64 ? 1 : 64 / P;
}
template <int N>
__global__ __launch_bounds__(zsolve_lines<N>() * (N / zsolve_elems<N>()) * MAXM, 3) void k_fftz_solve(ZArgs a, unsigned nlines) {
    constexpr int E = zsolve_elems<N>(), P = N / E, LPB = zsolve_lines<N>();
    constexpr int NCH = E == 16 ? 2 : 1;  // the solve buffer holds N/NCH modes per column at a time
    constexpr int CH = N / NCH, ECH = E / NCH;
    using LAY = LayContig<N>;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int c = threadIdx.x / P, j = threadIdx.x % P;
    const int f = c / LPB, l = c % LPB;
    const size_t line = (size_t)blockIdx.x * LPB + l;
    const bool ok = line < nlines;
    cd* __restrict__ Wl = a.W + f * a.fstride + line * N;
    const cd* __restrict__ Win = a.dup ? a.W + line * N : Wl;  // dup: every field starts from field 0's line
#ifndef SDC_ZSOLVE_NT
#define SDC_ZSOLVE_NT 0   // bit 0: nontemporal loads, bit 1: nontemporal stores of the lines
#endif
    cd r[E];
#pragma unroll
    for (int i = 0; i < E; ++i) {
#if SDC_ZSOLVE_NT & 1
        r[i] = ok ? cd{__builtin_nontemporal_load(&Win[j + i * P].x), __builtin_nontemporal_load(&Win[j + i * P].y)} : cd{0.0, 0.0};
#else
        r[i] = ok ? Win[j + i * P] : cd{0.0, 0.0};
#endif
    }
    fft_line<N, -1, LAY, (P <= 64), E>(r, j, c, lds, a.tw);
    __syncthreads();  // the solve buffer aliases other waves' exchange planes (and all loads of a shared line are done)

    cd* buf = reinterpret_cast<cd*>(lds);  // [column][CH]
    const int nthreads = a.nf * LPB * P;
#pragma unroll
    for (int ph = 0; ph < NCH; ++ph) {
#pragma unroll
        for (int i = 0; i < ECH; ++i) buf[c * CH + j + i * P] = r[ph * ECH + i];
        __syncthreads();
        for (int item = threadIdx.x; item < LPB * CH; item += nthreads) {
            const int ll = item / CH, kk = item % CH;
            const size_t ln = (size_t)blockIdx.x * LPB + ll;
            const int kz = ph * CH + kk;
            cd lam = a.lamI[kz], mu = cd{0.0, 0.0};
            if (a.lamE) mu = a.lamE[kz];
            if (a.ndim == 3) {
                const int kx = (int)(ln / N) % (N / 2 + 1), ky = (int)(ln % N);
                lam = cadd(lam, cadd(a.lamI[kx], a.lamI[ky]));
                if (a.lamE) mu = cadd(mu, cadd(a.lamE[kx], a.lamE[ky]));
            } else if (a.ndim == 2) {
                const int kx = (int)(ln % (N / 2 + 1));
                lam = cadd(lam, a.lamI[kx]);
                if (a.lamE) mu = cadd(mu, a.lamE[kx]);
            }
            cd u[MAXM];
#pragma unroll
            for (int m = 0; m < MAXM; ++m) {
                if (m < a.nf) {
                    cd acc = buf[(m * LPB + ll) * CH + kk];
                    if (a.coupled) {
#pragma unroll
                        for (int q = 0; q < m; ++q) {
                            const double ci = a.cI[m][q], ce = a.cE[m][q];
                            const cd coef = cd{ci * lam.x + ce * mu.x, ci * lam.y + ce * mu.y};
                            acc = cfma(coef, u[q], acc);
                        }
                    }
                    const double al = a.alpha[m];
                    // The operators applied to the SOLUTION act on the real field the inverse transform makes of it: its modes 0
                    // and N/2 are real (numpy's irfft drops their imaginary parts; a symbol with an odd derivative is not real at
                    // N/2 - AdvectionDiffusionEquation_1D_FFT.py:227-238 followed by :203-207).  1-D lines only (in more
                    // dimensions the caller evaluates through real space when the symbol is complex).
                    if (a.dup && !a.apply && m == 1 && a.ndim == 1 && (kz == 0 || 2 * kz == N)) u[0].y = 0.0;
                    if (a.dup && m == 1) u[m] = a.apply ? cmul(acc, mu) : cmul(u[0], lam);  // (every column of a dup launch holds field 0's line)
                    else if (a.dup && m == 2) u[m] = cmul(u[0], mu);
                    else u[m] = a.apply ? cmul(acc, lam) : cmul(acc, cinv_fast(cd{1.0 - al * lam.x, -al * lam.y}));
                    buf[(m * LPB + ll) * CH + kk] = cscale(u[m], a.invN);
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < ECH; ++i) r[ph * ECH + i] = buf[c * CH + j + i * P];
        __syncthreads();
    }
    // opaque copy of the lane index: without it the forward transform's twiddles stay live (~100 VGPRs)
    // through the whole kernel for reuse in the inverse
    int j2 = j;
    asm volatile("" : "+v"(j2));
    fft_line<N, +1, LAY, (P <= 64), E>(r, j2, c, lds, a.tw);
    if (ok) {
#pragma unroll
        for (int i = 0; i < E; ++i) {
#if SDC_ZSOLVE_NT & 2
            __builtin_nontemporal_store(r[i].x, &Wl[j2 + i * P].x);
            __builtin_nontemporal_store(r[i].y, &Wl[j2 + i * P].y);
#else
            Wl[j2 + i * P] = r[i];
#endif
        }
    }
}

// plain transform along the contiguous axis, src -> dst (may alias), optionally scaled: forward to bring u0 /
// node values into the fully transformed domain of the spectral cache, inverse after the spectral sweep
// SYM: the operator symbol (lamI [+ lamE]) is multiplied in before the transform (own instantiation: the plain pass keeps
// its code)
template <int N, int DIR, bool SYM = false>
__global__ __launch_bounds__(z_lines_per_block<N>() * (N / fft_elems(N)) * MAXM, 4) void k_fftz_plain(
    const cd* __restrict__ src, cd* __restrict__ dst, size_t fstride, const cd* __restrict__ tw, unsigned nlines,
    double scale, const cd* __restrict__ src_one = nullptr, int one = -1, const cd* __restrict__ lamI = nullptr,
    const cd* __restrict__ lamE = nullptr, int ndim = 0, const cd* sub = nullptr) {
    constexpr int E = fft_elems(N), P = N / E, LPB = z_lines_per_block<N>();
    using LAY = LayContig<N>;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int c = threadIdx.x / P, j = threadIdx.x % P;
    const int f = c / LPB, l = c % LPB;
    const size_t line = (size_t)blockIdx.x * LPB + l;
    const bool ok = line < nlines;
    const size_t base = f * fstride + line * N;
    cd r[E];
    // (source field `one` may live outside the strided block: the last node's spectrum, SpecArgs::SL)
#ifndef SDC_ZPLAIN_NT
#define SDC_ZPLAIN_NT 0   // bit 0: nontemporal loads, bit 1: nontemporal stores of the lines
#endif
    const cd* __restrict__ in = (f == one) ? src_one + line * N : src + base;
#pragma unroll
    for (int i = 0; i < E; ++i) {
#if SDC_ZPLAIN_NT & 1
        r[i] = ok ? cd{__builtin_nontemporal_load(&in[j + i * P].x), __builtin_nontemporal_load(&in[j + i * P].y)} : cd{0.0, 0.0};
#else
        r[i] = ok ? in[j + i * P] : cd{0.0, 0.0};
#endif
    }
    if (sub) {  // the difference of two spectra is transformed (sub may be the destination: every thread reads its own modes first)
#pragma unroll
        for (int i = 0; i < E; ++i)
            if (ok) r[i] = csub(r[i], sub[base + j + i * P]);
    }
    if (scale != 1.0) {
#pragma unroll
        for (int i = 0; i < E; ++i) r[i] = cscale(r[i], scale);
    }
    if constexpr (SYM) {
        // operator applied in Fourier space before the inverse pass: multiply mode (kx, ky, kz) by the sum of the 1-D
        // symbols of the implicit (and explicit) stencils - the spectrum of f(u) for the linear right-hand sides
        cd sxy = cd{0.0, 0.0};
        if (ndim == 3) {
            const int kx = (int)(line / N), ky = (int)(line % N);
            sxy = cadd(lamI[kx], lamI[ky]);
            if (lamE) sxy = cadd(sxy, cadd(lamE[kx], lamE[ky]));
        } else if (ndim == 2) {
            sxy = lamI[line];
            if (lamE) sxy = cadd(sxy, lamE[line]);
        }
#pragma unroll
        for (int i = 0; i < E; ++i) {
            const int kz = j + i * P;
            cd sym = cadd(sxy, lamI[kz]);
            if (lamE) sym = cadd(sym, lamE[kz]);
            r[i] = ok ? cmul(r[i], sym) : r[i];
        }
    }
    fft_line<N, DIR, LAY, (N / fft_elems(N)) <= 64>(r, j, c, lds, tw);
    if (ok) {
#pragma unroll
        for (int i = 0; i < E; ++i) {
#if SDC_ZPLAIN_NT & 2
            __builtin_nontemporal_store(r[i].x, &dst[base + j + i * P].x);
            __builtin_nontemporal_store(r[i].y, &dst[base + j + i * P].y);
#else
            dst[base + j + i * P] = r[i];
#endif
        }
    }
}

// field q of the spectral cache: the last one lives behind its own pointer (see SpecArgs::SL); q is a compile-time
// constant wherever this is used (unrolled loops), so the choice costs nothing at run time
#define SPEC_FIELD(a, q, NF_) ((q) == (NF_) - 1 ? (a).SL : (a).S + (size_t)(q) * (a).fstride)

// Sweep in the transformed domain (DESIGN.md "spectral reuse").  For linear f(u) = A u (+ B u) the gathered
// right-hand side of node m is  u0 + dt sum_j (Q-QI)[m][j] A u_j^k (+ explicit part): its transform follows
// from the transforms of u0 and of the previous iterate, which the previous sweep left in S.  One launch
// reads S0 and S[0..nf), applies gather + node-coupled solve per mode, writes the new spectra back to S and
// their inverse transform along the contiguous axis to W (input of the inverse y / x passes).
struct SpecArgs {
    cd* S;
    cd* SL;  // spectrum of the LAST node: S + (nf-1)*fstride, or the buffer it swaps with S0 from step to step (sdc_advance)
    size_t fstride;
    const cd* S0;
    cd* W;
    const cd *tw, *lamI, *lamE;
    double gI[MAXM][MAXM], gE[MAXM][MAXM];  // dt (Q - QI), dt (Q - QE), inner MxM blocks
    double cI[MAXM][MAXM], cE[MAXM][MAXM], alpha[MAXM];
    double rQ[MAXM][MAXM];  // dt Q, inner MxM block (RES)
    // u-independent forcing profile(x) * g(t) (heatNd_forced): its contribution to the right-hand side of node m and to
    // the residual is cP[m] * profile with cP[m] = dt sum_j Q[m][j] g(t_j); SP = transform of the profile, or null
    const cd* SP;
    double cP[MAXM];
    double invN;
    int nf, ndim, coupled, spread;
    int real_sym;  // the implicit symbol is real (symmetric stencil: heat): half the multiplications per node (wave-uniform branch)
    // Iterates that are not stored (DESIGN.md "sweeps recomputed from the start value"): after a spread predictor every
    // iterate of a linear problem is a function of the transform of u0 alone.  replay = number of earlier sweeps this
    // launch repeats in registers (same coefficients) before it does its own; virt: the new spectra are not stored
    // either (2: ... and the line transform gets the iterate itself instead of its residual); last_only: only the last
    // node's spectrum is stored.
    int replay, virt, last_only;
    unsigned block0;  // first workgroup of a launch that covers a range of lines
    // Table of the real node multipliers (mode pairs, long runs of sweeps): G[(line * NF + m) * (N/2 + 1) + p], p = 0 .. N/2.
    // gmode 1: the launch replays as usual and leaves the multipliers of ITS iterate in the table; 2: it takes the
    // multipliers of the previous iterate from the table, does one sweep on them and writes them back (no replay).
    double* G;
    int gmode;
    // mode-pair launches: the last node's spectrum of the NEW iterate goes to SL on the way (what k_spec_store_pairs would
    // write with last_only: the end value / next start value) - the launch has its multipliers in registers
    int store_last;
    // Trail (time-parallel levels): the iterate after nsw unstored sweeps depends on ns start values - src[0] the one the spread
    // predictor copied to every node, src[i] the i-th one received since; sweep s (0-based) started from src[vsrc[s]].
    const cd* src[MAXTRAIL];
    int ns, nsw;
    unsigned char vsrc[MAXVSWEEPS];
    int scnt[MAXTRAIL];   // sweeps that started from src[i] (consecutive: a start value only changes when one is received)
    unsigned long long scnt_packed;   // the same, 8 bits per source: the kernel's loop bounds without a load inside its loops
    double gIrow[MAXM];   // row sums of gI
    // the trail's z launch can transform one more line per z line on the way: src[ns-1] - src[ns-2], the difference of the
    // last two start values, which the put-off x pass of the PREVIOUS iterate's residual is waiting for (flush_x) - the
    // launch has both in registers; dz = where it goes (scaled like the residual lines), or null
    cd* dz;
};

// one thread per Fourier mode: gather on the cached transforms + node-coupled solve, S updated in place.
// RES: the transform of the collocation residual of the NEW iterate, u0 - u_m + dt sum_j Q[m][j] f(u_j)
// (core/sweeper.py:186-199 with f(u) = (A + B) u), goes to W[m]; its inverse transform only has to be reduced
// to a max norm, so neither U[1..M] nor F[1..M] are needed in real space to continue sweeping.
// acc + lam * t  and  acc / (1 - al * lam)  with the short forms for a real symbol (wave-uniform choice)
DEVI cd sym_fma(cd lam, cd t, cd acc, int real_sym) {
    if (real_sym) return cd{fma(lam.x, t.x, acc.x), fma(lam.x, t.y, acc.y)};
    return cfma(lam, t, acc);
}
DEVI cd node_divide(cd acc, cd lam, double al, int real_sym) {
    if (real_sym) {
        const double inv = fast_rcp(1.0 - al * lam.x);
        return cd{acc.x * inv, acc.y * inv};
    }
    return cmul(acc, cinv_fast(cd{1.0 - al * lam.x, -al * lam.y}));
}

// Iterates that are not stored.  From "all nodes equal u0" every iterate of a linear problem is u_m = g_m * u0 per mode,
// with node multipliers g_m that depend on the symbols only: a sweep maps them as
//   g_m <- (1 + lam sum_q gI[m][q] g_q(old) + lam sum_{q<m} cI[m][q] g_q(new) [+ mu (...)]) / (1 - alpha_m lam),
// real arithmetic for a real symbol without explicit part (heat): 9 fused multiply-adds per node and sweep on average
// instead of ~40 on the transformed values themselves.  The residual of the result is h_m * u0 with
//   h_m = 1 - g_m + (lam + mu) sum_j rQ[m][j] g_j.
// one sweep on the multipliers (inv[m] = 1 / (1 - alpha_m lam))
template <int NF>
DEVI void virt_step_real(const SpecArgs& a, double lam, const double (&inv)[NF], double (&g)[NF]) {
    double o[NF];
#pragma unroll
    for (int q = 0; q < NF; ++q) o[q] = g[q];
#pragma unroll
    for (int m = 0; m < NF; ++m) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < NF; ++q) t = fma(a.gI[m][q], o[q], t);
        if (a.coupled) {
#pragma unroll
            for (int q = 0; q < m; ++q) t = fma(a.cI[m][q], g[q], t);
        }
        g[m] = fma(lam, t, 1.0) * inv[m];
    }
}
template <int NF>
DEVI void virt_multipliers_real(const SpecArgs& a, double lam, int nsweeps, double (&g)[NF]) {
    double inv[NF];
#pragma unroll
    for (int m = 0; m < NF; ++m) {
        inv[m] = fast_rcp(1.0 - a.alpha[m] * lam);
        g[m] = 1.0;
    }
    for (int s = 0; s < nsweeps; ++s) virt_step_real<NF>(a, lam, inv, g);
}
// ... of the iterate whose multipliers the table holds: one more sweep on them
template <int NF>
DEVI void virt_multipliers_from_table(const SpecArgs& a, double lam, double (&g)[NF]) {
    double inv[NF];
#pragma unroll
    for (int m = 0; m < NF; ++m) inv[m] = fast_rcp(1.0 - a.alpha[m] * lam);
    virt_step_real<NF>(a, lam, inv, g);
}
template <int NF, bool HASE>
DEVI void virt_multipliers(const SpecArgs& a, cd lam, cd mu, int nsweeps, cd (&g)[NF]) {
    cd inv[NF];
#pragma unroll
    for (int m = 0; m < NF; ++m) {
        inv[m] = cinv_fast(cd{1.0 - a.alpha[m] * lam.x, -a.alpha[m] * lam.y});
        g[m] = cd{1.0, 0.0};
    }
    for (int s = 0; s < nsweeps; ++s) {
        cd o[NF];
#pragma unroll
        for (int q = 0; q < NF; ++q) o[q] = g[q];
#pragma unroll
        for (int m = 0; m < NF; ++m) {
            cd tI = cd{0.0, 0.0}, tE = cd{0.0, 0.0};
#pragma unroll
            for (int q = 0; q < NF; ++q) {
                tI = cd{fma(a.gI[m][q], o[q].x, tI.x), fma(a.gI[m][q], o[q].y, tI.y)};
                if (HASE) tE = cd{fma(a.gE[m][q], o[q].x, tE.x), fma(a.gE[m][q], o[q].y, tE.y)};
            }
            if (a.coupled) {
#pragma unroll
                for (int q = 0; q < m; ++q) {
                    tI = cd{fma(a.cI[m][q], g[q].x, tI.x), fma(a.cI[m][q], g[q].y, tI.y)};
                    if (HASE) tE = cd{fma(a.cE[m][q], g[q].x, tE.x), fma(a.cE[m][q], g[q].y, tE.y)};
                }
            }
            cd acc = cfma(lam, tI, cd{1.0, 0.0});
            if (HASE) acc = cfma(mu, tE, acc);
            g[m] = cmul(acc, inv[m]);
        }
    }
}
// Trail of unstored sweeps whose start values differ (time-parallel levels: u[0] is replaced between sweeps,
// controller_MPI.py:218-233).  One sweep with start value s maps the node vector as  u <- T u + b s  per mode, with
//   (T x)_m = lam (sum_q gI[m][q] x_q + sum_{q<m} cI[m][q] (T x)_q) / (1 - alpha_m lam),   b = the same with "1 +" for "gI x".
// From "all nodes equal src[0]" the iterate after J sweeps is  T^J 1 * src[0] + sum_s T^(J-1-s) b * src[vsrc[s]]:  two chains
// of real node multipliers (2J - 1 applications of T) whatever the number of start values.  Real symmetric symbol only.
template <int NF>
DEVI void trail_apply(const SpecArgs& a, double lam, const double (&inv)[NF], double (&x)[NF]) {
    double o[NF];
#pragma unroll
    for (int q = 0; q < NF; ++q) o[q] = x[q];
#pragma unroll
    for (int m = 0; m < NF; ++m) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < NF; ++q) t = fma(a.gI[m][q], o[q], t);
        if (a.coupled) {
#pragma unroll
            for (int q = 0; q < m; ++q) t = fma(a.cI[m][q], x[q], t);
        }
        x[m] = lam * t * inv[m];
    }
}
// node values of the modes p (lo) and n - p (hi; paired) of the line that starts at `base` after a.nsw sweeps
template <int NF>
DEVI void trail_iterate(const SpecArgs& a, double lam, size_t base, int p_, int n, bool paired, cd (&ulo)[NF], cd (&uhi)[NF]) {
    double inv[NF], x[NF], y[NF];
    const int J = a.nsw;
    // (the first start value's modes are on their way while the multipliers are set up)
    const cd* sp = a.src[a.vsrc[J - 1]];
    cd slo = sp[base + p_], shi = paired ? sp[base + n - p_] : cd{0.0, 0.0};
#pragma unroll
    for (int m = 0; m < NF; ++m) {
        inv[m] = fast_rcp(1.0 - a.alpha[m] * lam);
        y[m] = 1.0;
        ulo[m] = uhi[m] = cd{0.0, 0.0};
    }
#pragma unroll
    for (int m = 0; m < NF; ++m) {   // b
        double t = 0.0;
        if (a.coupled) {
#pragma unroll
            for (int q = 0; q < m; ++q) t = fma(a.cI[m][q], x[q], t);
        }
        x[m] = fma(lam, t, 1.0) * inv[m];
    }
    for (int s = 0; s < J; ++s) {
        // x = T^s b belongs to sweep J - 1 - s; the start value of the sweep before it is fetched one step ahead
        const cd clo = slo, chi = shi;
        const cd* sn = s + 1 < J ? a.src[a.vsrc[J - 2 - s]] : a.src[0];
        slo = sn[base + p_];
        if (paired) shi = sn[base + n - p_];
#pragma unroll
        for (int m = 0; m < NF; ++m) {
            ulo[m] = cd{fma(x[m], clo.x, ulo[m].x), fma(x[m], clo.y, ulo[m].y)};
            uhi[m] = cd{fma(x[m], chi.x, uhi[m].x), fma(x[m], chi.y, uhi[m].y)};
        }
        trail_apply<NF>(a, lam, inv, y);
        if (s + 1 < J) trail_apply<NF>(a, lam, inv, x);
    }
    // y = T^J 1: what is left of the spread predictor's copies of src[0]
#pragma unroll
    for (int m = 0; m < NF; ++m) {
        ulo[m] = cd{fma(y[m], slo.x, ulo[m].x), fma(y[m], slo.y, ulo[m].y)};
        uhi[m] = cd{fma(y[m], shi.x, uhi[m].x), fma(y[m], shi.y, uhi[m].y)};
    }
}
// The residual of that iterate against the current start value (the last sweep's), and the last node's value, for one mode
// pair.  Works on the real node multipliers alone: sweeps come in runs that share a start value (it changes only when one is
// received, so the sweeps of source i are consecutive: a.scnt[i] of them), the multipliers of a run are the sum of its chain
// vectors, and a start value's modes enter once, when its run is over:
//   R_m += (delta - G_m + lam sum_q rQ[m][q] G_q) * s,   u_M += G_M * s.
// All start values are fetched before the first multiplication (slo / shi, filled by the caller: the loads of a thread's NEXT
// pair are in flight while it works on this one).
#ifndef TRAIL_S
#define TRAIL_S 4
#endif
#define TRAIL_S_NOTE   // start values the in-register variant handles (sdc_set_timeslice_options' default); more: iterates are stored
// (x, y) <- (T x, T y): one pass over the coefficients for both chains.  Written column by column - every old value goes into
// all NF row sums before the next one is touched, and a finished new value into the rows below it at once - so that 2 NF sums
// advance side by side and the dependent part is two operations per row, not a row's whole sum (the launch has two waves per
// SIMD: its f64 pipe lives on the independent operations of ONE wave)
// cf: the sweep's coefficients in LDS - gI[NF][NF], cI[NF][NF], alpha[NF], row sums of gI[NF] - read with one address for the
// whole wave.  (As kernel arguments they do not fit the scalar registers beside everything else: the launch kept fetching them
// from the argument segment inside its loops, a wait of the whole wave each time - 7 of its 16 ms.)
template <int NF>
struct TrailCoef {
    static constexpr int GI = 0, CI = NF * NF, AL = 2 * NF * NF, GR = 2 * NF * NF + NF, COUNT = 2 * NF * NF + 2 * NF;
};
template <int NF>
DEVI void trail_apply2(const double* cf, bool coupled, double lam, const double (&inv)[NF], double (&x)[NF], double (&y)[NF]) {
    using TC = TrailCoef<NF>;
    double tx[NF], ty[NF];
#pragma unroll
    for (int m = 0; m < NF; ++m) tx[m] = ty[m] = 0.0;
#pragma unroll
    for (int q = 0; q < NF; ++q) {
#pragma unroll
        for (int m = 0; m < NF; ++m) {
            const double g = cf[TC::GI + m * NF + q];
            tx[m] = fma(g, x[q], tx[m]);
            ty[m] = fma(g, y[q], ty[m]);
        }
    }
#pragma unroll
    for (int q = 0; q < NF; ++q) {
        const double li = lam * inv[q];
        x[q] = li * tx[q];
        y[q] = li * ty[q];
        if (coupled) {
#pragma unroll
            for (int m = q + 1; m < NF; ++m) {
                const double cq = cf[TC::CI + m * NF + q];
                tx[m] = fma(cq, x[q], tx[m]);
                ty[m] = fma(cq, y[q], ty[m]);
            }
        }
    }
}
template <int NF>
DEVI void trail_residual(const SpecArgs& a, const double* cf, double lam, const cd (&slo)[TRAIL_S], const cd (&shi)[TRAIL_S],
                         cd (&rlo)[NF], cd (&rhi)[NF], cd& ulo, cd& uhi) {
    using TC = TrailCoef<NF>;
    const bool coupled = a.coupled != 0;
    // The residual of the NEW iterate is  lam dt (Q - QI)(u_new - u_old): the sweep's own equation (generic_implicit.py:75-103)
    // subtracted from the residual's (core/sweeper.py:186-199).  Per start value only the DIFFERENCE of its node multipliers
    // between the last two iterates is needed - D_i = x(end of its run) - x(end of the run before it) [+ T^(J-1)(T 1 - 1) for
    // src[0]]: two chain vectors - and the coefficients are the chain loop's own (no second matrix in scalar registers, no
    // cancellation of O(1) terms either).  The last node's value needs the last component of the multipliers themselves.
    double inv[NF], x[NF], z[NF], xp[NF];
    const int J = a.nsw;
    ulo = uhi = cd{0.0, 0.0};
#pragma unroll
    for (int m = 0; m < NF; ++m) {
        inv[m] = fast_rcp(1.0 - cf[TC::AL + m] * lam);
        rlo[m] = rhi[m] = cd{0.0, 0.0};
        xp[m] = 0.0;
    }
#pragma unroll
    for (int m = 0; m < NF; ++m) {   // x = b,  z = T 1 - 1
        double tb = 0.0, t1 = cf[TC::GR + m];
        if (coupled) {
#pragma unroll
            for (int q = 0; q < m; ++q) {
                const double cq = cf[TC::CI + m * NF + q];
                tb = fma(cq, x[q], tb);
                t1 = fma(cq, z[q] + 1.0, t1);
            }
        }
        x[m] = fma(lam, tb, 1.0) * inv[m];
        z[m] = fma(lam * t1, inv[m], -1.0);
    }
    double zsum = 1.0;   // last node of T^J 1 = 1 + sum_k (T^k z)_M
    int s = 0;
    // (ONE copy of the loop body whatever the number of start values: the launch is bound by its vector instructions, and
    // unrolled copies spill the coefficient matrix from scalar registers into vector lanes)
    const unsigned long long counts = a.scnt_packed;
    for (int i = a.ns - 1; i >= 0; --i) {   // the chain vector x = T^s b belongs to sweep J - 1 - s: the latest source first
        const int cnt = (int)((counts >> (8 * i)) & 255u);
        double gM = 0.0;
        for (int t = 0; t < cnt; ++t) {
            gM += x[NF - 1];
            zsum += z[NF - 1];
            if (t + 1 == cnt) {   // the run of src[i] ends with this chain vector
                cd sl = slo[0], sh = shi[0];
#pragma unroll
                for (int q = 1; q < TRAIL_S; ++q) {   // (selects on a wave-uniform condition: the arrays stay in registers)
                    const bool hit = i == q;
                    sl = cd{hit ? slo[q].x : sl.x, hit ? slo[q].y : sl.y};
                    sh = cd{hit ? shi[q].x : sh.x, hit ? shi[q].y : sh.y};
                }
                const double first = i == 0 ? 1.0 : 0.0;
                double D[NF];
#pragma unroll
                for (int m = 0; m < NF; ++m) {
                    D[m] = fma(first, z[m], x[m] - xp[m]);
                    xp[m] = x[m];
                }
                double tq[NF];
#pragma unroll
                for (int m = 0; m < NF; ++m) tq[m] = 0.0;
#pragma unroll
                for (int q = 0; q < NF; ++q)   // (column by column: NF sums side by side)
#pragma unroll
                    for (int m = 0; m < NF; ++m) tq[m] = fma(cf[TC::GI + m * NF + q], D[q], tq[m]);
#pragma unroll
                for (int m = 0; m < NF; ++m) {
                    const double h = lam * tq[m];
                    rlo[m] = cd{fma(h, sl.x, rlo[m].x), fma(h, sl.y, rlo[m].y)};
                    rhi[m] = cd{fma(h, sh.x, rhi[m].x), fma(h, sh.y, rhi[m].y)};
                }
                const double gl = fma(first, zsum, gM);
                ulo = cd{fma(gl, sl.x, ulo.x), fma(gl, sl.y, ulo.y)};
                uhi = cd{fma(gl, sh.x, uhi.x), fma(gl, sh.y, uhi.y)};
            }
            if (++s < J) trail_apply2<NF>(cf, coupled, lam, inv, x, z);
        }
    }
}
template <int NF>
DEVI void trail_fetch(const SpecArgs& a, size_t base, int p_, int n, bool paired, cd (&slo)[TRAIL_S], cd (&shi)[TRAIL_S]) {
    // (the partner of an unpaired mode is read from the mode itself and never used, the entries of a.src beyond the trail repeat
    // its last start value: no branch around a load - with one, the number of loads in flight is not known when the code is
    // made, and a wait for an earlier load waits for these as well)
    const size_t ilo = base + p_, ihi = base + (paired ? n - p_ : p_);
#pragma unroll
    for (int i = 0; i < TRAIL_S; ++i) {
        const cd* __restrict__ sp = a.src[i];
        slo[i] = sp[ilo];
        shi[i] = sp[ihi];
    }
}

// ... written out: all node spectra or (last_only) only the last one - what store_spectra does for a trail
template <int NF>
__global__ __launch_bounds__(256) void k_trail_store(SpecArgs a, int n, size_t nitems) {
    const int H = n / 2;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < nitems; idx += (size_t)gridDim.x * blockDim.x) {
        const int p_ = (int)(idx % (size_t)(H + 1));
        const size_t ln = idx / (size_t)(H + 1);
        double lxy = 0.0;
        if (a.ndim == 3) lxy = a.lamI[ln / n].x + a.lamI[ln % n].x;
        else if (a.ndim == 2) lxy = a.lamI[ln].x;
        const size_t base = ln * (size_t)n;
        const bool paired = p_ >= 1 && p_ < H;
        cd ulo[NF], uhi[NF];
        trail_iterate<NF>(a, a.lamI[p_].x + lxy, base, p_, n, paired, ulo, uhi);
#pragma unroll
        for (int m = 0; m < NF; ++m)
            if (!a.last_only || m == NF - 1) {
                SPEC_FIELD(a, m, NF)[base + p_] = ulo[m];
                if (paired) SPEC_FIELD(a, m, NF)[base + n - p_] = uhi[m];
            }
    }
}

// u[m] = g_m u0 (and, RES, r[m] = h_m u0) for the iterate after nsweeps sweeps
template <int NF, bool HASE, bool RES>
DEVI void virt_iterate(const SpecArgs& a, cd lam, cd mu, cd u0h, int nsweeps, cd (&u)[NF], cd (&r)[NF]) {
    if (!HASE && a.real_sym) {
        double g[NF];
        virt_multipliers_real<NF>(a, lam.x, nsweeps, g);
#pragma unroll
        for (int m = 0; m < NF; ++m) u[m] = cd{g[m] * u0h.x, g[m] * u0h.y};
        if constexpr (RES) {
#pragma unroll
            for (int m = 0; m < NF; ++m) {
                double t = 0.0;
#pragma unroll
                for (int q = 0; q < NF; ++q) t = fma(a.rQ[m][q], g[q], t);
                const double h = fma(lam.x, t, 1.0 - g[m]);
                r[m] = cd{h * u0h.x, h * u0h.y};
            }
        }
    } else {
        cd g[NF];
        virt_multipliers<NF, HASE>(a, lam, mu, nsweeps, g);
#pragma unroll
        for (int m = 0; m < NF; ++m) u[m] = cmul(g[m], u0h);
        if constexpr (RES) {
            const cd sym = HASE ? cadd(lam, mu) : lam;
#pragma unroll
            for (int m = 0; m < NF; ++m) {
                cd t = cd{0.0, 0.0};
#pragma unroll
                for (int q = 0; q < NF; ++q) t = cd{fma(a.rQ[m][q], g[q].x, t.x), fma(a.rQ[m][q], g[q].y, t.y)};
                const cd h = cfma(sym, t, cd{1.0 - g[m].x, -g[m].y});
                r[m] = cmul(h, u0h);
            }
        }
    }
}

template <int NF, bool RES>
__global__ __launch_bounds__(256) void k_spec_point(SpecArgs a, int n, size_t nmodes) {
    for (size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x; g < nmodes; g += (size_t)gridDim.x * blockDim.x) {
        const int kz = (int)(g % n);
        const size_t ln = g / n;
        cd lam = a.lamI[kz], mu = cd{0.0, 0.0};
        if (a.lamE) mu = a.lamE[kz];
        if (a.ndim == 3) {
            const int kx = (int)(ln / n), ky = (int)(ln % n);
            lam = cadd(lam, cadd(a.lamI[kx], a.lamI[ky]));
            if (a.lamE) mu = cadd(mu, cadd(a.lamE[kx], a.lamE[ky]));
        } else if (a.ndim == 2) {
            lam = cadd(lam, a.lamI[ln]);
            if (a.lamE) mu = cadd(mu, a.lamE[ln]);
        }
        const cd u0h = a.S0[g];
        const cd ph = a.SP ? a.SP[g] : cd{0.0, 0.0};
        cd old[NF], u[NF];
#pragma unroll
        for (int q = 0; q < NF; ++q) old[q] = a.spread ? u0h : SPEC_FIELD(a, q, NF)[g];
#pragma unroll
        for (int m = 0; m < NF; ++m) {
            cd acc = cd{fma(a.cP[m], ph.x, u0h.x), fma(a.cP[m], ph.y, u0h.y)};
            // real-weighted node sums first, one multiplication by each symbol afterwards (as in k_spec_z)
            cd tI = cd{0.0, 0.0}, tE = cd{0.0, 0.0};
#pragma unroll
            for (int q = 0; q < NF; ++q) {
                tI = cd{fma(a.gI[m][q], old[q].x, tI.x), fma(a.gI[m][q], old[q].y, tI.y)};
                tE = cd{fma(a.gE[m][q], old[q].x, tE.x), fma(a.gE[m][q], old[q].y, tE.y)};
            }
            if (a.coupled) {
#pragma unroll
                for (int q = 0; q < m; ++q) {
                    tI = cd{fma(a.cI[m][q], u[q].x, tI.x), fma(a.cI[m][q], u[q].y, tI.y)};
                    tE = cd{fma(a.cE[m][q], u[q].x, tE.x), fma(a.cE[m][q], u[q].y, tE.y)};
                }
            }
            acc = sym_fma(lam, tI, acc, a.real_sym);
            if (a.lamE) acc = cfma(mu, tE, acc);
            u[m] = node_divide(acc, lam, a.alpha[m], a.real_sym);
            SPEC_FIELD(a, m, NF)[g] = u[m];  // (nontemporal stores measured slower here: 18.5 vs 17.4 ms at 1024^3)
        }
        if constexpr (RES) {
            const cd sym = cadd(lam, mu);
#pragma unroll
            for (int m = 0; m < NF; ++m) {
                cd acc = csub(u0h, u[m]);
                acc = cd{fma(a.cP[m], ph.x, acc.x), fma(a.cP[m], ph.y, acc.y)};
                cd tR = cd{0.0, 0.0};
#pragma unroll
                for (int q = 0; q < NF; ++q) tR = cd{fma(a.rQ[m][q], u[q].x, tR.x), fma(a.rQ[m][q], u[q].y, tR.y)};
                acc = sym_fma(sym, tR, acc, a.real_sym && !a.lamE);
                a.W[m * a.fstride + g] = acc;
            }
        }
    }
}

// the spectra of an iterate that was never stored (store_spectra): S[m] = g_m * S0 after nsweeps sweeps, all nodes or
// (last_only) only the last one
template <int NF>
__global__ __launch_bounds__(256) void k_spec_store(SpecArgs a, int n, size_t nmodes, int nsweeps) {
    const bool p2 = (n & (n - 1)) == 0;    // (lines of 3 * 2^p / 5 * 2^p modes divide; the rest shifts)
    const int lg = 31 - __builtin_clz(n);
    for (size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x; g < nmodes; g += (size_t)gridDim.x * blockDim.x) {
        const int kz = p2 ? (int)(g & (size_t)(n - 1)) : (int)(g % (size_t)n);
        const size_t ln = p2 ? g >> lg : g / (size_t)n;
        cd lam = a.lamI[kz], mu = cd{0.0, 0.0};
        if (a.lamE) mu = a.lamE[kz];
        if (a.ndim == 3) {
            const int kx = p2 ? (int)(ln >> lg) : (int)(ln / (size_t)n), ky = p2 ? (int)(ln & (size_t)(n - 1)) : (int)(ln % (size_t)n);
            lam = cadd(lam, cadd(a.lamI[kx], a.lamI[ky]));
            if (a.lamE) mu = cadd(mu, cadd(a.lamE[kx], a.lamE[ky]));
        } else if (a.ndim == 2) {
            lam = cadd(lam, a.lamI[ln]);
            if (a.lamE) mu = cadd(mu, a.lamE[ln]);
        }
        const cd u0h = a.S0[g];
        cd u[NF], r[NF];
        if (a.lamE) virt_iterate<NF, true, false>(a, lam, mu, u0h, nsweeps, u, r);
        else virt_iterate<NF, false, false>(a, lam, mu, u0h, nsweeps, u, r);
#pragma unroll
        for (int m = 0; m < NF; ++m)
            if (!a.last_only || m == NF - 1) SPEC_FIELD(a, m, NF)[g] = u[m];
    }
}

// ... for a real symmetric symbol (heat): the modes kz and n - kz of a line share lam and hence the multipliers.  One
// thread per pair (p, n - p), p in [0, n/2); the thread with p = 0 also takes the unpaired mode n/2.
template <int NF>
__global__ __launch_bounds__(256) void k_spec_store_pairs(SpecArgs a, int n, size_t npairs, int nsweeps) {
    const bool p2 = (n & (n - 1)) == 0;
    const int lg = 31 - __builtin_clz(n);
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < npairs; idx += (size_t)gridDim.x * blockDim.x) {
        const int p_ = p2 ? (int)(idx & (size_t)(n / 2 - 1)) : (int)(idx % (size_t)(n / 2));
        const size_t ln = p2 ? idx >> (lg - 1) : idx / (size_t)(n / 2);
        double lxy = 0.0;
        if (a.ndim == 3) lxy = p2 ? a.lamI[ln >> lg].x + a.lamI[ln & (size_t)(n - 1)].x : a.lamI[ln / (size_t)n].x + a.lamI[ln % (size_t)n].x;
        else if (a.ndim == 2) lxy = a.lamI[ln].x;
        const size_t base = ln * (size_t)n;
        const size_t ilo = base + p_, ihi = base + (p_ ? n - p_ : n / 2);
        const cd lo = a.S0[ilo], hi = a.S0[ihi];
        double g[NF];
        // (gmode 2: the table holds the multipliers of exactly this iterate)
        const double* gt = a.gmode == 2 ? a.G + ln * (size_t)NF * (size_t)(n / 2 + 1) : nullptr;
        if (gt) {
#pragma unroll
            for (int m = 0; m < NF; ++m) g[m] = (!a.last_only || m == NF - 1) ? gt[(size_t)m * (n / 2 + 1) + p_] : 0.0;
        } else {
            virt_multipliers_real<NF>(a, a.lamI[p_].x + lxy, nsweeps, g);
        }
#pragma unroll
        for (int m = 0; m < NF; ++m)
            if (!a.last_only || m == NF - 1) SPEC_FIELD(a, m, NF)[ilo] = cd{g[m] * lo.x, g[m] * lo.y};
        if (p_ == 0) {
            if (gt) {
#pragma unroll
                for (int m = 0; m < NF; ++m) g[m] = (!a.last_only || m == NF - 1) ? gt[(size_t)m * (n / 2 + 1) + n / 2] : 0.0;
            } else {
                virt_multipliers_real<NF>(a, a.lamI[n / 2].x + lxy, nsweeps, g);
            }
        }
#pragma unroll
        for (int m = 0; m < NF; ++m)
            if (!a.last_only || m == NF - 1) SPEC_FIELD(a, m, NF)[ihi] = cd{g[m] * hi.x, g[m] * hi.y};
    }
}

// transform of the collocation residual of the CACHED iterate against the current S0 (u[0] was replaced after
// the sweep, e.g. by a receive): W[m] = S0 - S[m] + dt sum_j Q[m][j] (lam + mu) S[j]
template <int NF>
__global__ __launch_bounds__(256) void k_spec_residual(SpecArgs a, int n, size_t nmodes) {
    for (size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x; g < nmodes; g += (size_t)gridDim.x * blockDim.x) {
        const int kz = (int)(g % n);
        const size_t ln = g / n;
        cd sym = a.lamI[kz];
        if (a.lamE) sym = cadd(sym, a.lamE[kz]);
        if (a.ndim == 3) {
            const int kx = (int)(ln / n), ky = (int)(ln % n);
            sym = cadd(sym, cadd(a.lamI[kx], a.lamI[ky]));
            if (a.lamE) sym = cadd(sym, cadd(a.lamE[kx], a.lamE[ky]));
        } else if (a.ndim == 2) {
            sym = cadd(sym, a.lamI[ln]);
            if (a.lamE) sym = cadd(sym, a.lamE[ln]);
        }
        const cd u0h = a.S0[g];
        const cd ph = a.SP ? a.SP[g] : cd{0.0, 0.0};
        cd u[NF];
#pragma unroll
        for (int q = 0; q < NF; ++q) u[q] = SPEC_FIELD(a, q, NF)[g];
#pragma unroll
        for (int m = 0; m < NF; ++m) {
            cd acc = csub(u0h, u[m]);
            acc = cd{fma(a.cP[m], ph.x, acc.x), fma(a.cP[m], ph.y, acc.y)};
            cd tR = cd{0.0, 0.0};
#pragma unroll
            for (int q = 0; q < NF; ++q) tR = cd{fma(a.rQ[m][q], u[q].x, tR.x), fma(a.rQ[m][q], u[q].y, tR.y)};
            acc = cfma(sym, tR, acc);
            a.W[m * a.fstride + g] = acc;
        }
    }
}


#ifndef SDC_SPECZ_NT
#define SDC_SPECZ_NT 3  // bit 0: nontemporal stores of the new spectra, bit 1: of the transformed lines (1024^3: 24.0 -> 23.4 ms)
#endif
#ifndef SDC_SPECZ_WAVES
#define SDC_SPECZ_WAVES 4
#endif
#ifndef SDC_SPECZ_E256
#define SDC_SPECZ_E256 8
#endif
#ifndef SDC_SPECZ_E1024
#define SDC_SPECZ_E1024 16
#endif
#ifndef SDC_SPECZ_CH
#define SDC_SPECZ_CH 512
#endif
// Spectral sweep fused with the first inverse pass.  One workgroup owns the same LPB lines of all NF fields
// as k_fftz_plain does (column c = f*LPB + l on threads [c*P, (c+1)*P)), i.e. a contiguous span of LPB*N
// modes per field.  Phase 1 treats that span pointwise (one thread = one mode, all NF node values in
// registers: gather on the cached transforms, node-coupled solve, S updated in place, residual spectrum) and
// hands the field to transform - the residual spectrum (RES) or the new iterate - to the FFT threads through
// LDS in chunks of <= 512 modes; phase 2 is the inverse line transform, whose exchange planes reuse that LDS.
// Saves writing and re-reading NF spectra between k_spec_point and k_fftz_plain.
// MODE 0: sweep, the new iterate is transformed; 1: sweep, the residual spectrum is transformed (RES);
// 2: no sweep - residual spectrum of the CACHED iterate against the current S0 (u[0] was replaced).
// 3: like 1, but the iterate is a function of S0 alone (spread predictor, then a.replay sweeps that were never stored):
// reads S0 only, repeats those sweeps in registers, does its own, stores NOTHING but the transformed residual lines.
// Elements per thread of the line transform inside the fused kernel: a 512-line uses 8, so that its 64 threads are one
// wavefront and a workgroup owns ONE line per field like at 1024 (no second chunk whose work keeps the first chunk's
// elements alive in registers).
#ifndef SDC_SPECZ_VE1024
#define SDC_SPECZ_VE1024 16  // ... of the launch that recomputes the iterate (MODE 3)
#endif
#ifndef SDC_SPECZ_VWAVES
#define SDC_SPECZ_VWAVES SDC_SPECZ_WAVES
#endif
#ifndef SDC_SPECZ_VHOIST
#define SDC_SPECZ_VHOIST 1  // MODE 3: the S0 loads of all chunks are issued before the first chunk is worked on
#endif
template <int N, bool V = false>
constexpr int specz_elems() {
    return N == 512 ? 8 : (N == 256 ? SDC_SPECZ_E256 : (N == 1024 ? (V ? SDC_SPECZ_VE1024 : SDC_SPECZ_E1024) : fft_elems(N)));
}
template <int N, bool V = false>
constexpr int specz_lines() {
    constexpr int P = N / specz_elems<N, V>();
    return P >= 64 ? 1 : 64 / P;
}

template <int N, int NF, bool V>
constexpr int specz_threads() { return specz_lines<N, V>() * (N / specz_elems<N, V>()) * NF; }
// modes handed through LDS at a time: the largest divisor of the workgroup's span (lines x N) that is no more than SDC_SPECZ_CH
// (512 for the powers of two; 384 for spans of 768 = lines of 3 * 2^p, 320 for spans of 1280 = lines of 5 * 2^p)
constexpr int specz_chunk(int span) {
    int ch = span > SDC_SPECZ_CH ? SDC_SPECZ_CH : span;
    while (span % ch) --ch;
    return ch;
}
template <int N, bool V>
constexpr int specz_min_waves() { return specz_lines<N, V>() > 1 ? 2 : (V ? SDC_SPECZ_VWAVES : SDC_SPECZ_WAVES); }
// EXPL 0: no explicit part, 1: explicit stencil (symbol lamE), 2: u-independent forcing (profile spectrum SP).
// threads of the trail launch (k_trail_z): the node multipliers of the N/2 mode pairs of a line are the long part of that
// launch, one pair per thread wants more threads than the NF transforms have (8 waves instead of 5 at 1024 x 5: the
// waves beyond the transforms' only work on multipliers)
#ifndef SDC_TRAIL_WAVES
#define SDC_TRAIL_WAVES 2   // waves per SIMD the trail launch is compiled for (registers: 256 / 170 / 128)
#endif
template <int N, int NF>
constexpr int trail_threads() {
    constexpr int nt = specz_threads<N, NF, true>();
    constexpr int want = ((N / 2 + 63) / 64) * 64 > 512 ? 512 : ((N / 2 + 63) / 64) * 64;
    return (specz_lines<N, true>() == 1 && want > nt) ? want : nt;
}
template <int N, int NF, int MODE, int EXPL>
// (several lines per workgroup, N < 1024: the elements a thread takes from the first chunk stay live while it works on
// the second one - 2 waves / SIMD worth of registers instead of spilling; measured 5.7 -> 4.2 ms at 512^3)
__global__ __launch_bounds__((specz_threads<N, NF, (MODE >= 3)>()), (specz_min_waves<N, (MODE >= 3)>()))
void k_spec_z(SpecArgs a, unsigned nlines) {
    constexpr bool RES = MODE >= 1, UPD = MODE <= 1 || MODE >= 3, VIRT = MODE >= 3, PAIR = MODE >= 4,
                   GTAB = MODE == 5, HASE = EXPL == 1, HASP = EXPL == 2;
    static_assert(!(VIRT && HASP), "a forced iterate is not a function of the start value alone");
    constexpr int E = specz_elems<N, (MODE >= 3)>(), P = N / E, LPB = specz_lines<N, (MODE >= 3)>();
    static_assert(!PAIR || (LPB == 1 && !HASE), "mode pairs: one line per field and workgroup, real symbol only");
    constexpr int SPAN = LPB * N, CH = specz_chunk(SPAN), NCH = SPAN / CH, NT = LPB * P * NF;
    constexpr int ITS = (CH + NT - 1) / NT;  // modes per thread and chunk
    using LAY = LayContig<N>;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    cd* rbuf = reinterpret_cast<cd*>(lds);  // [NF][CH]
#ifndef SDC_Z_SWZ
#define SDC_Z_SWZ 0  // 1: every XCD a contiguous eighth of the lines of the launch
#endif
    const unsigned bsw = (SDC_Z_SWZ && (gridDim.x & 7u) == 0) ? (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    const unsigned bid = bsw + a.block0;  // (block0: a launch that covers only a range of lines, e.g. a group of kx planes)
    const int c = threadIdx.x / P, j = threadIdx.x % P;
    const int f = c / LPB, l = c % LPB;
    const size_t line = (size_t)bid * LPB + l;
    const bool ok = line < nlines;
    const size_t span0 = (size_t)bid * SPAN, nmodes = (size_t)nlines * N;
    cd r[E];
    if constexpr (PAIR) {
        // Real symmetric symbol: the modes kz and N - kz of a line share lam, hence the node multipliers.  One item
        // p in [0, N/2] per thread and round: multipliers once, both modes scaled by them.  (LPB == 1: the workgroup's
        // span is ONE line, kx and ky are the same for all of it.)
        constexpr int NI = N / 2 + 1, IT2 = (NI + NT - 1) / NT;
        const size_t base = (size_t)bid * N;
        cd lo[IT2], hi[IT2];
        double lz[IT2];
#pragma unroll
        for (int it = 0; it < IT2; ++it) {
            const int p_ = threadIdx.x + it * NT;
            lo[it] = hi[it] = cd{0.0, 0.0};
            lz[it] = 0.0;
#ifndef SDC_SPECZ_LD_NT
#define SDC_SPECZ_LD_NT 0   // nontemporal loads of the start-value modes (read once per launch)
#endif
            if (p_ < NI && ok) {
#if SDC_SPECZ_LD_NT
                lo[it] = cd{__builtin_nontemporal_load(&a.S0[base + p_].x), __builtin_nontemporal_load(&a.S0[base + p_].y)};
                if (p_ >= 1 && p_ < N / 2) hi[it] = cd{__builtin_nontemporal_load(&a.S0[base + N - p_].x), __builtin_nontemporal_load(&a.S0[base + N - p_].y)};
#else
                lo[it] = a.S0[base + p_];
                if (p_ >= 1 && p_ < N / 2) hi[it] = a.S0[base + N - p_];
#endif
                lz[it] = a.lamI[p_].x;
            }
        }
        double lxy = 0.0;
        if (a.ndim == 3) lxy = a.lamI[bid / N].x + a.lamI[bid % N].x;
        else if (a.ndim == 2) lxy = a.lamI[bid].x;
        double hm[IT2][NF];
#pragma unroll
        for (int it = 0; it < IT2; ++it) {
            const double lam = lz[it] + lxy;
            double g[NF];
            if constexpr (GTAB) {
                // long runs of sweeps: the multipliers of the previous iterate come from the table (gmode 2) instead of
                // being recomputed from 1 by a.replay sweeps, and those of this one go back into it
                const int p_ = threadIdx.x + it * NT;
                double* gp = a.G + (size_t)bid * NF * NI + p_;
                const bool mine = p_ < NI && ok;
                if (a.gmode == 2) {
#pragma unroll
                    for (int m = 0; m < NF; ++m) g[m] = mine ? gp[(size_t)m * NI] : 1.0;
                    virt_multipliers_from_table<NF>(a, lam, g);
                } else {
                    virt_multipliers_real<NF>(a, lam, a.replay + 1, g);
                }
                if (mine) {
#pragma unroll
                    for (int m = 0; m < NF; ++m) gp[(size_t)m * NI] = g[m];
                }
            } else {
                virt_multipliers_real<NF>(a, lam, a.replay + 1, g);
            }
            if (a.store_last) {
                const int p_ = threadIdx.x + it * NT;
                if (p_ < NI && ok) {
                    a.SL[base + p_] = cd{g[NF - 1] * lo[it].x, g[NF - 1] * lo[it].y};
                    if (p_ >= 1 && p_ < N / 2) a.SL[base + N - p_] = cd{g[NF - 1] * hi[it].x, g[NF - 1] * hi[it].y};
                }
            }
            if (a.virt == 2) {  // the iterate itself is wanted in real space (node values stored by every sweep)
#pragma unroll
                for (int m = 0; m < NF; ++m) hm[it][m] = g[m] * a.invN;
            } else {
#pragma unroll
                for (int m = 0; m < NF; ++m) {
                    double t = 0.0;
#pragma unroll
                    for (int q = 0; q < NF; ++q) t = fma(a.rQ[m][q], g[q], t);
                    hm[it][m] = fma(lam, t, 1.0 - g[m]) * a.invN;
                }
            }
        }
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
#pragma unroll
            for (int it = 0; it < IT2; ++it) {
                const int p_ = threadIdx.x + it * NT;
                if (p_ < NI) {
                    if (p_ / CH == ch) {
#pragma unroll
                        for (int m = 0; m < NF; ++m) rbuf[m * CH + (p_ % CH)] = cscale(lo[it], hm[it][m]);
                    }
                    const int q_ = N - p_;
                    if (p_ >= 1 && p_ < N / 2 && q_ / CH == ch) {
#pragma unroll
                        for (int m = 0; m < NF; ++m) rbuf[m * CH + (q_ % CH)] = cscale(hi[it], hm[it][m]);
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < E; ++i) {
                const int o = j + i * P;
                // (unconditional: a workgroup beyond the last line - there is none with one line per workgroup - would transform
                // what the buffer holds and store nothing; a test per element costs two register moves and a branch each)
                if (o / CH == ch) r[i] = rbuf[f * CH + (o % CH)];
            }
            __syncthreads();
        }
    }
    constexpr bool HOIST = VIRT && !PAIR && SDC_SPECZ_VHOIST;
    cd in0all[HOIST ? NCH : 1][ITS];
    if constexpr (HOIST) {
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
            for (int it = 0; it < ITS; ++it) {
                const int k = threadIdx.x + it * NT;
                const size_t g = span0 + (size_t)ch * CH + k;
                if (k < CH && g < nmodes) in0all[ch][it] = a.S0[g];
            }
    }
#pragma unroll
    for (int ch = 0; ch < (PAIR ? 0 : NCH); ++ch) {
        // all loads of this chunk first: (NF + 1) * ITS independent 16-byte loads per thread in flight
        cd in0[ITS], inq[ITS][VIRT ? 1 : NF], inp[ITS];
#pragma unroll
        for (int it = 0; it < ITS; ++it) {
            const int k = threadIdx.x + it * NT;
            const size_t g = span0 + (size_t)ch * CH + k;
            if (HASP) inp[it] = cd{0.0, 0.0};
            if constexpr (HOIST) {
                in0[it] = in0all[ch][it];
            } else if (k < CH && g < nmodes) {
                in0[it] = a.S0[g];
                if (HASP) inp[it] = a.SP[g];
                if constexpr (!VIRT) {
                    if (!UPD || !a.spread) {
#pragma unroll
                        for (int q = 0; q < NF; ++q) inq[it][q] = SPEC_FIELD(a, q, NF)[g];
                    }
                }
            }
        }
#pragma unroll
        for (int it = 0; it < ITS; ++it) {
            const int k = threadIdx.x + it * NT;
            const size_t g = span0 + (size_t)ch * CH + k;
            if (k < CH && g < nmodes) {
                const int kz = (int)(g % N);
                const size_t ln = g / N;
                cd lam = a.lamI[kz], mu = cd{0.0, 0.0};
                if (HASE) mu = a.lamE[kz];
                if (a.ndim == 3) {
                    const int kx = (int)(ln / N), ky = (int)(ln % N);
                    lam = cadd(lam, cadd(a.lamI[kx], a.lamI[ky]));
                    if (HASE) mu = cadd(mu, cadd(a.lamE[kx], a.lamE[ky]));
                } else if (a.ndim == 2) {
                    lam = cadd(lam, a.lamI[ln]);
                    if (HASE) mu = cadd(mu, a.lamE[ln]);
                }
                const cd u0h = in0[it];
                cd old[NF], u[NF];
                if constexpr (VIRT) {
                    cd rr[NF];
                    virt_iterate<NF, HASE, true>(a, lam, mu, u0h, a.replay + 1, u, rr);
#pragma unroll
                    for (int m = 0; m < NF; ++m) rbuf[m * CH + k] = cscale(rr[m], a.invN);
                }
#pragma unroll
                for (int q = 0; q < NF; ++q)
                    if constexpr (!VIRT) old[q] = (UPD && a.spread) ? u0h : inq[it][q];
#pragma unroll
                for (int m = 0; m < NF; ++m) {
                    if constexpr (VIRT) continue;
                    if constexpr (!UPD) {
                        u[m] = old[m];
                        continue;
                    }
                    cd acc = u0h;
                    if (HASP) acc = cd{fma(a.cP[m], inp[it].x, u0h.x), fma(a.cP[m], inp[it].y, u0h.y)};
                    // the symbols do not depend on the node: form the real-weighted node sums first and multiply by
                    // lam (mu) once - 2 fused multiply-adds per term instead of 6 operations
                    cd tI = cd{0.0, 0.0}, tE = cd{0.0, 0.0};
#pragma unroll
                    for (int q = 0; q < NF; ++q) {
                        tI = cd{fma(a.gI[m][q], old[q].x, tI.x), fma(a.gI[m][q], old[q].y, tI.y)};
                        if (HASE) tE = cd{fma(a.gE[m][q], old[q].x, tE.x), fma(a.gE[m][q], old[q].y, tE.y)};
                    }
                    if (a.coupled) {
#pragma unroll
                        for (int q = 0; q < m; ++q) {
                            tI = cd{fma(a.cI[m][q], u[q].x, tI.x), fma(a.cI[m][q], u[q].y, tI.y)};
                            if (HASE) tE = cd{fma(a.cE[m][q], u[q].x, tE.x), fma(a.cE[m][q], u[q].y, tE.y)};
                        }
                    }
                    acc = sym_fma(lam, tI, acc, a.real_sym);
                    if (HASE) acc = cfma(mu, tE, acc);
                    u[m] = node_divide(acc, lam, a.alpha[m], a.real_sym);
#if SDC_SPECZ_NT & 1
                    __builtin_nontemporal_store(u[m].x, &SPEC_FIELD(a, m, NF)[g].x);
                    __builtin_nontemporal_store(u[m].y, &SPEC_FIELD(a, m, NF)[g].y);
#else
                    SPEC_FIELD(a, m, NF)[g] = u[m];
#endif
                }
                if constexpr (VIRT) {
                } else if constexpr (RES) {
                    const cd sym = HASE ? cadd(lam, mu) : lam;
#pragma unroll
                    for (int m = 0; m < NF; ++m) {
                        cd acc = csub(u0h, u[m]);
                        if (HASP) acc = cd{fma(a.cP[m], inp[it].x, acc.x), fma(a.cP[m], inp[it].y, acc.y)};
                        cd tR = cd{0.0, 0.0};
#pragma unroll
                        for (int q = 0; q < NF; ++q) tR = cd{fma(a.rQ[m][q], u[q].x, tR.x), fma(a.rQ[m][q], u[q].y, tR.y)};
                        acc = sym_fma(sym, tR, acc, a.real_sym && !HASE);
                        rbuf[m * CH + k] = cscale(acc, a.invN);
                    }
                } else {
#pragma unroll
                    for (int m = 0; m < NF; ++m) rbuf[m * CH + k] = cscale(u[m], a.invN);
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < E; ++i) {
            const int o = l * N + j + i * P;  // offset of element i inside the span
            if (o / CH == ch) r[i] = ok ? rbuf[f * CH + (o % CH)] : cd{0.0, 0.0};
        }
        __syncthreads();  // the next chunk / the exchange planes of the transform overwrite the buffer
    }
    fft_line<N, +1, LAY, P <= 64, E>(r, j, c, lds, a.tw);
    if (ok) {
        cd* __restrict__ dst = a.W + f * a.fstride + line * N;
#pragma unroll
        for (int i = 0; i < E; ++i) {
#if SDC_SPECZ_NT & 2
            __builtin_nontemporal_store(r[i].x, &dst[j + i * P].x);
            __builtin_nontemporal_store(r[i].y, &dst[j + i * P].y);
#else
            dst[j + i * P] = r[i];
#endif
        }
    }
}

// The recomputing z launch of a slice on the trail as a PERSISTENT launch: one workgroup per CU (the hand-over buffer of a line
// of every field fills half the LDS, its registers the other limit) walks over lines; the start values of its next line are on
// their way while it works on the multipliers of this one, and the stores of a line drain under the next line's arithmetic.
// With one line per workgroup and launch the CU's memory pipes idle while the multipliers are made and its ALUs while the line
// is fetched and stored: 16.9 - 27.9 ms for 1 - 4 replayed sweeps at 1024^3 x 5, of which 8.1 - 12.8 are memory time.
// The Nyquist mode of every line (item N/2 + 1 of N/2 + 1: a second round of one lane) is made by a launch of its own
// (k_trail_nyq) and added to the hand-over buffer from a table.
template <int NF, bool DZ>
__global__ __launch_bounds__(256) void k_trail_nyq(SpecArgs a, int n, unsigned nlines, cd* __restrict__ nyq) {
    using TC = TrailCoef<NF>;
    __shared__ double cf[TC::COUNT];
    if (threadIdx.x < NF * NF) {
        cf[TC::GI + threadIdx.x] = a.gI[threadIdx.x / NF][threadIdx.x % NF];
        cf[TC::CI + threadIdx.x] = a.cI[threadIdx.x / NF][threadIdx.x % NF];
    }
    if (threadIdx.x < NF) {
        cf[TC::AL + threadIdx.x] = a.alpha[threadIdx.x];
        cf[TC::GR + threadIdx.x] = a.gIrow[threadIdx.x];
    }
    __syncthreads();
    const unsigned line = blockIdx.x * blockDim.x + threadIdx.x;
    if (line >= nlines) return;
    const int H = n / 2;
    double lxy = 0.0;
    if (a.ndim == 3) lxy = a.lamI[line / n].x + a.lamI[line % n].x;
    else if (a.ndim == 2) lxy = a.lamI[line].x;
    const size_t base = (size_t)line * n;
    cd slo[TRAIL_S], shi[TRAIL_S];
    trail_fetch<NF>(a, base, H, n, false, slo, shi);
    cd rlo[NF], rhi[NF], ulo, uhi;
    trail_residual<NF>(a, cf, a.lamI[H].x + lxy, slo, shi, rlo, rhi, ulo, uhi);
    if (a.store_last) a.SL[base + H] = ulo;
#pragma unroll
    for (int m = 0; m < NF; ++m) nyq[(size_t)line * (NF + DZ) + m] = cscale(rlo[m], a.invN);
    if constexpr (DZ) nyq[(size_t)line * (NF + 1) + NF] = cscale(csub(a.src[a.ns - 1][base + H], a.src[a.ns - 2][base + H]), a.invN);
}

template <int N, int NF, bool DZ>
__global__ __launch_bounds__((trail_threads<N, NF>()), SDC_TRAIL_WAVES)
void k_trail_z(SpecArgs a, unsigned nlines, const cd* __restrict__ nyq) {
    constexpr int NL = NF + (DZ ? 1 : 0);   // lines handed to the transforms: the residual of every node [+ the difference line]
    constexpr int E = specz_elems<N, true>(), P = N / E, NTT = trail_threads<N, NF>(), H = N / 2;
    static_assert(specz_lines<N, true>() == 1 && P <= 64 && H <= NTT, "one line per workgroup, one mode pair per thread");
    using LAY = LayContig<N>;
    using TC = TrailCoef<NF>;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    cd* rbuf = reinterpret_cast<cd*>(lds);           // [NL][N]: the residual lines handed over to the transforms
    double* cf = lds + 2 * (size_t)NL * N;           // the sweep's coefficients (see TrailCoef)
    // the twiddle factors the line transform looks up, in LDS: with one workgroup per CU nobody else's work hides the latency of
    // a table look-up in L2, and the transform has two dependent ones per line (1.5 of 6.7 us per line at 1024^3)
    cd* twl = reinterpret_cast<cd*>(cf + ((TC::COUNT + 1) & ~1));
    static_assert(LAY::doubles(NL) <= 2 * NL * N && NL * (N / specz_elems<N, true>()) <= trail_threads<N, NF>(),
                  "the exchange planes of the transform alias the hand-over buffer; one wave per line");
    for (int i = threadIdx.x; i < N / 2; i += NTT) twl[i] = a.tw[i];
    if (threadIdx.x < NF * NF) {
        cf[TC::GI + threadIdx.x] = a.gI[threadIdx.x / NF][threadIdx.x % NF];
        cf[TC::CI + threadIdx.x] = a.cI[threadIdx.x / NF][threadIdx.x % NF];
    }
    if (threadIdx.x < NF) {
        cf[TC::AL + threadIdx.x] = a.alpha[threadIdx.x];
        cf[TC::GR + threadIdx.x] = a.gIrow[threadIdx.x];
    }
    __syncthreads();
    const int c = threadIdx.x / P, j = threadIdx.x % P;   // (c < NF: this wave transforms field c)
    const bool mine = (int)threadIdx.x < H;
    const int p_ = mine ? (int)threadIdx.x : H - 1;       // (a lane without a pair follows along on a valid one, stores nothing)
    const bool paired = p_ >= 1;
    const double lz = a.lamI[p_].x;
    cd slo[2][TRAIL_S], shi[2][TRAIL_S];                  // (two sets: the line being worked on, the line being fetched)
    unsigned line = blockIdx.x;
    // (what a line needs beside its start values - the symbol of its kx, ky, its Nyquist entries - is asked for one line ahead
    // and BEFORE the start values of the line after: a wait for these small loads then does not wait for the big ones)
    // (no branches around these loads: the kx / ky entries of the symbol table exist for every line index below nlines)
    const double wu = a.ndim >= 2 ? 1.0 : 0.0, wv = a.ndim == 3 ? 1.0 : 0.0;
    const bool d3 = a.ndim == 3, d2 = a.ndim == 2;
    const unsigned last = nlines - 1;
    auto iu_of = [=](unsigned ln) { const unsigned lc = ln < last ? ln : last; return d3 ? lc / N : (d2 ? lc : 0u); };
    auto iv_of = [=](unsigned ln) { const unsigned lc = ln < last ? ln : last; return d3 ? lc % N : 0u; };
    auto nyq_of = [=](unsigned ln) {
        const unsigned lc = ln < last ? ln : last;
        return nyq[(size_t)lc * NL + ((int)threadIdx.x < NL ? threadIdx.x : 0)];
    };
    double lu = a.lamI[iu_of(line)].x, lv = a.lamI[iv_of(line)].x;
    cd nq = nyq_of(line);
    if (line < nlines) trail_fetch<NF>(a, (size_t)line * N, p_, N, paired, slo[0], shi[0]);
    auto step = [&](auto CUR) {
        constexpr int cur = decltype(CUR)::value, nxt = cur ^ 1;
        const size_t base = (size_t)line * N;
        const unsigned nl = line + gridDim.x;
        const double lxy = wu * lu + wv * lv;
        const cd nq_cur = nq;
        lu = a.lamI[iu_of(nl)].x;
        lv = a.lamI[iv_of(nl)].x;
        nq = nyq_of(nl);
        trail_fetch<NF>(a, (size_t)(nl < last ? nl : last) * N, p_, N, paired, slo[nxt], shi[nxt]);   // (past the end: the last line again)
        cd rlo[NF], rhi[NF], ulo, uhi;
        trail_residual<NF>(a, cf, lz + lxy, slo[cur], shi[cur], rlo, rhi, ulo, uhi);
        __syncthreads();   // the transforms of the previous line are done with their exchange planes (first line: cf is there)
        if (mine) {
            if (a.store_last) {   // the last node's spectrum: what the wire carries, and the next step's start value
                a.SL[base + p_] = ulo;
                if (paired) a.SL[base + N - p_] = uhi;
            }
#pragma unroll
            for (int m = 0; m < NF; ++m) {
                rbuf[m * N + p_] = cscale(rlo[m], a.invN);
                if (paired) rbuf[m * N + N - p_] = cscale(rhi[m], a.invN);
            }
        }
        if constexpr (DZ) {   // the difference of the last two start values: one more line
            cd dl = slo[cur][0], dh = shi[cur][0], el = slo[cur][0], eh = shi[cur][0];
#pragma unroll
            for (int q = 1; q < TRAIL_S; ++q) {   // (selects on wave-uniform conditions: the arrays stay in registers)
                const bool hn = a.ns - 1 == q, ho = a.ns - 2 == q;
                dl = cd{hn ? slo[cur][q].x : dl.x, hn ? slo[cur][q].y : dl.y};
                dh = cd{hn ? shi[cur][q].x : dh.x, hn ? shi[cur][q].y : dh.y};
                el = cd{ho ? slo[cur][q].x : el.x, ho ? slo[cur][q].y : el.y};
                eh = cd{ho ? shi[cur][q].x : eh.x, ho ? shi[cur][q].y : eh.y};
            }
            if (mine) {
                rbuf[NF * N + p_] = cscale(csub(dl, el), a.invN);
                if (paired) rbuf[NF * N + N - p_] = cscale(csub(dh, eh), a.invN);
            }
        }
        if ((int)threadIdx.x < NL) rbuf[threadIdx.x * N + H] = nq_cur;
        __syncthreads();
        cd r[E];
        if (c < NL) {
#pragma unroll
            for (int i = 0; i < E; ++i) r[i] = rbuf[c * N + j + i * P];
        }
        __syncthreads();   // (every transforming wave has read its column: the exchange planes may overwrite the buffer)
        if (c < NL) {      // (the line transforms synchronise inside their own waves, P <= 64: the other waves go on)
            fft_line<N, +1, LAY, true, E>(r, j, c, lds, twl);
            cd* __restrict__ dst = (DZ && c == NF) ? a.dz + base : a.W + c * a.fstride + base;
#pragma unroll
            for (int i = 0; i < E; ++i) dst[j + i * P] = r[i];
        }
        line = nl;
    };
    while (line < nlines) {
        step(std::integral_constant<int, 0>{});
        if (line >= nlines) break;
        step(std::integral_constant<int, 1>{});
    }
}
