// In-register / LDS Stockham FFT building blocks for f64 complex lines of length N = 2^p (2 <= N <= 2048) or
// N = 3 * 2^p (24 .. 768: the even grids the reference accepts that are not powers of two, generic_ND_FD.py:126-129) on
// gfx950.  One line is spread over P = N/E threads with E = min(16, N) elements per thread (12 for the lengths with a factor
// 3: radix-4 / radix-2 stages, one radix-3 stage LAST, so that the number of finished sub-transforms NS stays a power of
// two where an index is reduced by it); element i of thread j is line index j + i*P.  Every stage reads that same strided set (so loads/stores to global memory
// keep one pattern for all N) and scatters through LDS to the Stockham output position.  Real and imaginary
// planes go through LDS one after the other, which halves the LDS footprint (N*8 bytes per line).
#pragma once
#include <hip/hip_runtime.h>

typedef double2 cd;
#define DEVI __device__ __forceinline__

// streaming accesses of the in-place middle-axis pass (every element is touched once): nontemporal loads / stores
// measured 3 % faster there at 1024^3; slower for the norm pass; the fused sweep kernel has its own switch (SDC_SPECZ_NT)
#ifndef SDC_NT
#define SDC_NT 1
#endif
DEVI cd ld_stream(const cd* p) {
#if SDC_NT
    return cd{__builtin_nontemporal_load(&p->x), __builtin_nontemporal_load(&p->y)};
#else
    return *p;
#endif
}
DEVI void st_stream(cd* p, cd v) {
#if SDC_NT
    __builtin_nontemporal_store(v.x, &p->x);
    __builtin_nontemporal_store(v.y, &p->y);
#else
    *p = v;
#endif
}
DEVI cd cadd(cd a, cd b) { return cd{a.x + b.x, a.y + b.y}; }
DEVI cd csub(cd a, cd b) { return cd{a.x - b.x, a.y - b.y}; }
DEVI cd cmul(cd a, cd b) { return cd{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
DEVI cd cscale(cd a, double s) { return cd{a.x * s, a.y * s}; }
DEVI cd cconj(cd a) { return cd{a.x, -a.y}; }
DEVI cd cfma(cd a, cd b, cd c) {  // a*b + c
    return cd{fma(a.x, b.x, fma(-a.y, b.y, c.x)), fma(a.x, b.y, fma(a.y, b.x, c.y))};
}
DEVI cd cinv(cd d) {  // 1/d
    double m = 1.0 / (d.x * d.x + d.y * d.y);
    return cd{d.x * m, -d.y * m};
}

// DIR = -1: forward transform exp(-2 pi i jk/N); DIR = +1: inverse (unnormalised)
template <int DIR>
DEVI cd mul_dir_i(cd a) {  // a * (DIR * i)
    return DIR < 0 ? cd{a.y, -a.x} : cd{-a.y, a.x};
}
template <int DIR>
DEVI cd tw_dir(cd w) {  // tables hold the forward twiddle
    return DIR < 0 ? w : cd{w.x, -w.y};
}

// cos(2 pi m/16), sin(2 pi m/16), m = 0..7
__device__ static const double kC16[8] = {1.0,
                                          0.92387953251128673848,
                                          0.70710678118654752440,
                                          0.38268343236508977173,
                                          0.0,
                                          -0.38268343236508977173,
                                          -0.70710678118654752440,
                                          -0.92387953251128673848};
__device__ static const double kS16[8] = {0.0,
                                          0.38268343236508977173,
                                          0.70710678118654752440,
                                          0.92387953251128673848,
                                          1.0,
                                          0.92387953251128673848,
                                          0.70710678118654752440,
                                          0.38268343236508977173};

template <int DIR, int M16>
DEVI cd mul_w16(cd a) {  // a * W16^(M16) in direction DIR, 0 <= M16 < 16
    constexpr int m = M16 & 15;
    if constexpr (m == 0) return a;
    else if constexpr (m == 4) return mul_dir_i<DIR>(a);
    else if constexpr (m == 8) return cd{-a.x, -a.y};
    else if constexpr (m == 12) return mul_dir_i<-DIR>(a);
    else if constexpr (m > 8) { cd b = mul_w16<DIR, m - 8>(a); return cd{-b.x, -b.y}; }
    else {
        const double c = kC16[m], s = (DIR < 0 ? -kS16[m] : kS16[m]);
        return cd{a.x * c - a.y * s, a.x * s + a.y * c};
    }
}

template <int DIR>
DEVI void bf2(cd& a, cd& b) {
    cd t = a;
    a = cadd(t, b);
    b = csub(t, b);
}
// 3-point DFT: X1 = x0 - (x1 + x2)/2 + DIR i (sqrt 3 / 2)(x1 - x2), X2 its mirror
template <int DIR>
DEVI void bf3(cd& a0, cd& a1, cd& a2) {
    constexpr double h = 0.86602540378443864676;  // sqrt(3) / 2
    const cd s = cadd(a1, a2), d = mul_dir_i<DIR>(cscale(csub(a1, a2), h));
    const cd m = cd{a0.x - 0.5 * s.x, a0.y - 0.5 * s.y};
    a0 = cadd(a0, s);
    a1 = cadd(m, d);
    a2 = csub(m, d);
}
// 5-point DFT (Winograd-style: two mirrored pairs): with s1 = x1 + x4, s2 = x2 + x3, d1 = x1 - x4, d2 = x2 - x3
//   X1,4 = x0 + c1 s1 + c2 s2  +- DIR i (S1 d1 + S2 d2),   X2,3 = x0 + c2 s1 + c1 s2  +- DIR i (S2 d1 - S1 d2)
// c_k = cos(2 pi k/5), S_k = sin(2 pi k/5)
template <int DIR>
DEVI void bf5(cd& a0, cd& a1, cd& a2, cd& a3, cd& a4) {
    constexpr double c1 = 0.30901699437494742410, c2 = -0.80901699437494742410;
    constexpr double S1 = 0.95105651629515357212, S2 = 0.58778525229247312917;
    const cd s1 = cadd(a1, a4), s2 = cadd(a2, a3), d1 = csub(a1, a4), d2 = csub(a2, a3);
    const cd m1 = cd{fma(c1, s1.x, fma(c2, s2.x, a0.x)), fma(c1, s1.y, fma(c2, s2.y, a0.y))};
    const cd m2 = cd{fma(c2, s1.x, fma(c1, s2.x, a0.x)), fma(c2, s1.y, fma(c1, s2.y, a0.y))};
    const cd e1 = mul_dir_i<DIR>(cd{fma(S1, d1.x, S2 * d2.x), fma(S1, d1.y, S2 * d2.y)});
    const cd e2 = mul_dir_i<DIR>(cd{fma(S2, d1.x, -S1 * d2.x), fma(S2, d1.y, -S1 * d2.y)});
    a0 = cadd(a0, cadd(s1, s2));
    a1 = cadd(m1, e1);
    a4 = csub(m1, e1);
    a2 = cadd(m2, e2);
    a3 = csub(m2, e2);
}
template <int DIR>
DEVI void bf4(cd& a0, cd& a1, cd& a2, cd& a3) {
    cd t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), t3 = mul_dir_i<DIR>(csub(a1, a3));
    a0 = cadd(t0, t2);
    a1 = cadd(t1, t3);
    a2 = csub(t0, t2);
    a3 = csub(t1, t3);
}

// R-point DFT of x[0..R-1] (natural order in, natural order out)
template <int R, int DIR>
DEVI void dft(cd (&x)[R]) {
    if constexpr (R == 2) {
        bf2<DIR>(x[0], x[1]);
    } else if constexpr (R == 3) {
        bf3<DIR>(x[0], x[1], x[2]);
    } else if constexpr (R == 4) {
        bf4<DIR>(x[0], x[1], x[2], x[3]);
    } else if constexpr (R == 5) {
        bf5<DIR>(x[0], x[1], x[2], x[3], x[4]);
    } else if constexpr (R == 8) {
        // 8 = 4 (n1) x 2 (n2): x[2*n1 + n2]
        bf4<DIR>(x[0], x[2], x[4], x[6]);
        bf4<DIR>(x[1], x[3], x[5], x[7]);  // A[k1][n2] at x[2*k1 + n2]
        x[3] = mul_w16<DIR, 2>(x[3]);
        x[5] = mul_w16<DIR, 4>(x[5]);
        x[7] = mul_w16<DIR, 6>(x[7]);
        bf2<DIR>(x[0], x[1]);
        bf2<DIR>(x[2], x[3]);
        bf2<DIR>(x[4], x[5]);
        bf2<DIR>(x[6], x[7]);  // X[k1 + 4*k2] at x[2*k1 + k2]
        cd y[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) y[k] = x[2 * (k & 3) + (k >> 2)];
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = y[k];
    } else {
        static_assert(R == 16, "radix");
        // 16 = 4 (n1) x 4 (n2): x[4*n1 + n2]
        bf4<DIR>(x[0], x[4], x[8], x[12]);
        bf4<DIR>(x[1], x[5], x[9], x[13]);
        bf4<DIR>(x[2], x[6], x[10], x[14]);
        bf4<DIR>(x[3], x[7], x[11], x[15]);  // A[k1][n2] at x[4*k1 + n2]
        x[5] = mul_w16<DIR, 1>(x[5]);
        x[6] = mul_w16<DIR, 2>(x[6]);
        x[7] = mul_w16<DIR, 3>(x[7]);
        x[9] = mul_w16<DIR, 2>(x[9]);
        x[10] = mul_w16<DIR, 4>(x[10]);
        x[11] = mul_w16<DIR, 6>(x[11]);
        x[13] = mul_w16<DIR, 3>(x[13]);
        x[14] = mul_w16<DIR, 6>(x[14]);
        x[15] = mul_w16<DIR, 9>(x[15]);
        bf4<DIR>(x[0], x[1], x[2], x[3]);
        bf4<DIR>(x[4], x[5], x[6], x[7]);
        bf4<DIR>(x[8], x[9], x[10], x[11]);
        bf4<DIR>(x[12], x[13], x[14], x[15]);  // X[k1 + 4*k2] at x[4*k1 + k2]
        cd y[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) y[k] = x[4 * (k & 3) + (k >> 2)];
#pragma unroll
        for (int k = 0; k < 16; ++k) x[k] = y[k];
    }
}

#ifndef SDC_FFT_E
#define SDC_FFT_E 16
#endif
constexpr bool fft_has3(int N) { return N % 3 == 0; }
constexpr bool fft_has5(int N) { return N % 5 == 0; }
constexpr int fft_odd(int N) { return fft_has3(N) ? 3 : (fft_has5(N) ? 5 : 1); }   // the ONE odd factor a length may have
constexpr int fft_elems(int N) {
    return fft_has3(N) ? (N < 12 ? N : 12) : (fft_has5(N) ? (N < 20 ? N : 20) : (N < SDC_FFT_E ? N : SDC_FFT_E));
}
// lengths the line transforms handle: 2^p, 3 * 2^p from 24 on and 5 * 2^p from 40 on (the thread counts of the kernels want
// P = N / 12 resp. N / 20 = 2^q >= 2)
constexpr bool fft_length_ok(int N) {
    if (N % 15 == 0) return false;
    int m = N / fft_odd(N);
    return N >= 2 && (m & (m - 1)) == 0 && (N % 3 != 0 || N >= 24) && (N % 5 != 0 || N >= 40);
}

// LDS index maps (in doubles).  The skew (pos >> 4) breaks the power-of-two strides of the stage-1 scatter.
// kUnit: doubles between neighbouring (skewed) positions of one column.  A run of positions pos0 + m * d with d a multiple of
// 16 - or d = 1 inside one group of 16 - is affine in m: idx(col, pos0) + m * kUnit * (d + d / 16) resp. + m * kUnit, which lets
// the exchange address all its elements from ONE computed index plus compile-time offsets (SDC_FFT_AFFINE).
template <int N>
struct LayContig {  // lines are separate: [col][pos]
    static constexpr int kLine = N + (N >> 4);
    static constexpr int kUnit = 1;
    DEVI static int idx(int col, int pos) { return col * kLine + pos + (pos >> 4); }
    static constexpr int doubles(int cols) { return cols * kLine; }
};
#ifndef SDC_LAYCOLS_PAD
#define SDC_LAYCOLS_PAD 4
#endif
template <int N>
struct LayCols {  // like LayContig, with the columns 8 dwords apart modulo the 64 banks: the LDS serves 16 lanes (128 bytes) per
                  // cycle, and the 16 lanes of a group hold 8 columns x 2 consecutive rows when the unpacked line is written -
                  // 8 x 4 dwords side by side.  (16 dwords apart, rounds 2 - 5: columns c and c + 4 shared their banks - the
                  // norm-only x pass kept the CU's LDS pipe busy 96 % of its time, a third of that in conflicts: round 6)
    static constexpr int kLine = N + (N >> 4) + SDC_LAYCOLS_PAD;
    static constexpr int kUnit = 1;
    DEVI static int idx(int col, int pos) { return col * kLine + pos + (pos >> 4); }
    static constexpr int doubles(int cols) { return cols * kLine; }
};
template <int N, int T>
struct LayStrided {  // T columns interleaved: [pos][col]
    static constexpr int kUnit = T;
    DEVI static int idx(int col, int pos) { return (pos + (pos >> 4)) * T + col; }
    static constexpr int doubles(int) { return (N + (N >> 4)) * T; }
};
#ifndef SDC_FFT_AFFINE
#define SDC_FFT_AFFINE 1
#endif

// one Stockham stage on the registers of one thread: radix R, NS = product of the previous radices
template <int N, int R, int NS, int DIR, int EE = fft_elems(N)>
DEVI void fft_butterflies(cd (&r)[EE], int j, const cd* __restrict__ tw) {
    constexpr int E = EE, P = N / E, NB = E / R;
#pragma unroll
    for (int q = 0; q < NB; ++q) {
        cd x[R];
#pragma unroll
        for (int t = 0; t < R; ++t) x[t] = r[q + t * NB];
        if constexpr (NS > 1) {
            // w^t, t = 1..R-1, from ONE table load: powers by squaring / products of depth <= 4, so the
            // rounding error stays at a few ulp while R-2 dependent L1/L2 look-ups disappear
            const int k = (j + q * P) & (NS - 1);
            constexpr int step = N / (NS * R);
            const cd w1 = tw_dir<DIR>(tw[k * step]);
            x[1] = cmul(x[1], w1);
            if constexpr (R == 3) {
                x[2] = cmul(x[2], cmul(w1, w1));
            } else if constexpr (R > 2) {
                const cd w2 = cmul(w1, w1);
                cd run[4] = {cmul(w2, w2), w1, w2, cmul(w2, w1)};  // w^4, w^1, w^2, w^3
                x[2] = cmul(x[2], run[2]);
                x[3] = cmul(x[3], run[3]);
                if constexpr (R > 4) {
                    const cd w4 = run[0];
                    x[4] = cmul(x[4], w4);
#pragma unroll
                    for (int t = 5; t < R; ++t) {  // w^t = w^(t-4) * w^4: four short chains, five live powers
                        run[t & 3] = cmul(run[t & 3], w4);
                        x[t] = cmul(x[t], run[t & 3]);
                    }
                }
            }
        }
        dft<R, DIR>(x);
#pragma unroll
        for (int t = 0; t < R; ++t) r[q + t * NB] = x[t];
    }
}

// block-wide barrier, or - when every column lives inside one wavefront - only a compiler-level fence: the
// lanes of a wave run in lock step and the LDS serves one wave's requests in order
template <bool WAVE>
DEVI void fft_sync() {
    if constexpr (WAVE) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
}

// scatter the stage output through LDS and read back the strided set of the next stage
template <int N, int R, int NS, class LAY, bool WAVE = false, int EE = fft_elems(N)>
DEVI void fft_exchange(cd (&r)[EE], int j, int col, double* lds) {
    constexpr int E = EE, P = N / E, NB = E / R;
    // positions written: base + t * NS; read: j + i * P.  Affine in t / i when the steps are multiples of 16 (no carry into the
    // skew) or, for NS = 1, when base is a multiple of 16 and t < 16
    constexpr bool WAFF = SDC_FFT_AFFINE && (NS % 16 == 0 || (NS == 1 && R == 16));
    constexpr bool RAFF = SDC_FFT_AFFINE && P % 16 == 0;
    constexpr int WSTEP = LAY::kUnit * (NS % 16 == 0 ? NS + NS / 16 : 1), RSTEP = LAY::kUnit * (P + P / 16);
    int wbase[NB];
#pragma unroll
    for (int q = 0; q < NB; ++q) {
        const int b = j + q * P;
        const int k = b & (NS - 1);
        wbase[q] = WAFF ? LAY::idx(col, (b - k) * R + k) : (b - k) * R + k;
    }
    const int rbase = LAY::idx(col, j);
#pragma unroll
    for (int part = 0; part < 2; ++part) {
#pragma unroll
        for (int q = 0; q < NB; ++q) {
#pragma unroll
            for (int t = 0; t < R; ++t) {
                const double v = part == 0 ? r[q + t * NB].x : r[q + t * NB].y;
                if constexpr (WAFF) lds[wbase[q] + t * WSTEP] = v;
                else lds[LAY::idx(col, wbase[q] + t * NS)] = v;
            }
        }
        fft_sync<WAVE>();
#pragma unroll
        for (int i = 0; i < E; ++i) {
            const double v = RAFF ? lds[rbase + i * RSTEP] : lds[LAY::idx(col, j + i * P)];
            if (part == 0) r[i].x = v;
            else r[i].y = v;
        }
        fft_sync<WAVE>();
    }
}

template <int N, int NS, int DIR, class LAY, bool WAVE, int EE = fft_elems(N)>
DEVI void fft_stages(cd (&r)[EE], int j, int col, double* lds, const cd* __restrict__ tw) {
    constexpr int REM = N / NS;
    constexpr int RMAX = EE < 16 ? EE : 16;  // a butterfly cannot be wider than the elements a thread holds
    // lengths with a factor 3 or 5 (12 / 20 elements per thread): radix 4 while the power-of-two part lasts, a radix 2 if one
    // is left, the odd radix last
    constexpr int ODD = fft_odd(N);
    constexpr int REM2 = REM / ODD;
    constexpr int R = ODD > 1 ? (REM2 >= 4 ? 4 : (REM2 == 2 ? 2 : ODD)) : (REM >= RMAX ? RMAX : REM);
    static_assert(EE % R == 0, "elements per thread must hold whole butterflies");
    fft_butterflies<N, R, NS, DIR, EE>(r, j, tw);
    if constexpr (NS * R < N) {
        fft_exchange<N, R, NS, LAY, WAVE, EE>(r, j, col, lds);
        fft_stages<N, NS * R, DIR, LAY, WAVE, EE>(r, j, col, lds, tw);
    }
}

// Full length-N transform of the line held as r[i] <-> index j + i*P; result in the same arrangement.
// All threads of the block must call this (it contains barriers when N > 16).  WAVE = true: the P threads
// of a column sit in ONE wavefront (P <= 64, columns wave-aligned), so no block barrier is needed.
// EE: elements per thread (default min(16, N)); EE = 8 spreads a 512-line over a whole wavefront.
template <int N, int DIR, class LAY, bool WAVE = false, int EE = fft_elems(N)>
DEVI void fft_line(cd (&r)[EE], int j, int col, double* lds, const cd* __restrict__ tw) {
    fft_stages<N, 1, DIR, LAY, WAVE, EE>(r, j, col, lds, tw);
}
