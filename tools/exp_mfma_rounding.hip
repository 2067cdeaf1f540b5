// How does v_mfma_f32_32x32x16_f16 round?  One MFMA D = A*B + C on random fp16 operands and an fp32 C, against the
// exactly rounded result (products of fp16 are exact in double; 16 of them + C fit double's 53 bits for the ranges used).
// Reports, in ulps of the exact result: signed mean (a bias means truncation somewhere), rms, max, and the fraction of
// correctly rounded results -- for C of the size of the dot product (a GEMM chain mid-way) and for C much larger than the
// products (the late part of a long chain), plus a few hand-made cases (ties, sub-half-ulp products that only count
// together).  Then the same for a CHAIN of N MFMAs against (a) the exact sum rounded once, (b) a chain of exactly
// rounded steps.  build: hipcc --offload-arch=gfx950 -O2 tools/exp_mfma_rounding.hip -o /tmp/exp_mfma_rounding
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

// A [32][16] row-major, B [16][32] row-major (k, col), C/D [32][32]; nchain MFMAs: A, B advance by one tile each
__global__ void mfma_chain(const _Float16* A, const _Float16* B, const float* C, float* D, int nchain) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    f32x16 acc;
    for (int q = 0; q < 16; ++q) acc[q] = C[((q & 3) + 8 * (q >> 2) + 4 * h) * 32 + r];
    for (int n = 0; n < nchain; ++n) {
        f16x8 a, b;
        for (int j = 0; j < 8; ++j) {
            a[j] = A[(size_t)n * 512 + r * 16 + 8 * h + j];
            b[j] = B[(size_t)n * 512 + (8 * h + j) * 32 + r];
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    }
    for (int q = 0; q < 16; ++q) D[((q & 3) + 8 * (q >> 2) + 4 * h) * 32 + r] = acc[q];
}

static double ulp_of(double x) {
    int e;
    frexp(fabs(x), &e);          // |x| = m * 2^e, m in [0.5, 1)
    return ldexp(1.0, e - 24);
}

static double urand() { return (double)rand() / RAND_MAX; }

struct Stats { double mean = 0, sq = 0, mx = 0; long n = 0, exact = 0; };
static void add(Stats& s, double got, double want_exact) {
    const double u = ulp_of(want_exact);
    const double e = (got - want_exact) / u;
    s.mean += e; s.sq += e * e; s.mx = fmax(s.mx, fabs(e)); s.n++;
    if ((float)want_exact == (float)got) s.exact++;
}
static void show(const char* name, const Stats& s) {
    printf("%-58s mean %+.4f rms %.4f max %.3f ulp, correctly rounded %.4f  (n = %ld)\n", name, s.mean / s.n, sqrt(s.sq / s.n),
           s.mx, (double)s.exact / s.n, s.n);
}

int main() {
    const int NCH = 64;
    _Float16 *dA, *dB;
    float *dC, *dD;
    hipMalloc(&dA, NCH * 512 * 2); hipMalloc(&dB, NCH * 512 * 2); hipMalloc(&dC, 4096); hipMalloc(&dD, 4096);
    std::vector<_Float16> A(NCH * 512), B(NCH * 512);
    std::vector<float> C(1024), D(1024);
    auto run = [&](int nchain) {
        hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(dC, C.data(), 4096, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(mfma_chain, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD, nchain);
        hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
    };
    auto exact_step = [&](int n, int i, int j, double c) {      // c + sum_k A[n][i][k] B[n][k][j] in double
        double s = c;
        for (int k = 0; k < 16; ++k) s += (double)(float)A[n * 512 + i * 16 + k] * (double)(float)B[n * 512 + k * 32 + j];
        return s;
    };
    srand(1);
    // ---- single MFMA, random data ----
    for (int mode = 0; mode < 4; ++mode) {
        // mode 0: C ~ dot product size; 1: C = 64 x; 2: C = 4096 x; 3: C = 0
        const double cscale[4] = {4.0, 256.0, 16384.0, 0.0};
        Stats s;
        for (int rep = 0; rep < 200; ++rep) {
            for (auto& x : A) x = (_Float16)(float)(urand() * 2 - 1);
            for (auto& x : B) x = (_Float16)(float)(urand() * 2 - 1);
            for (auto& x : C) x = (float)((urand() * 2 - 1) * cscale[mode]);
            run(1);
            for (int i = 0; i < 32; ++i)
                for (int j = 0; j < 32; ++j) add(s, D[i * 32 + j], exact_step(0, i, j, C[i * 32 + j]));
        }
        const char* names[4] = {"one MFMA, |C| ~ |dot|", "one MFMA, |C| ~ 64 |dot|", "one MFMA, |C| ~ 4096 |dot|", "one MFMA, C = 0"};
        show(names[mode], s);
    }
    // ---- hand-made: C = 1, products of 2^-26 each (1/8 ulp of 1.0 = 2^-23): only together do they count ----
    for (int np = 1; np <= 16; ++np) {
        for (auto& x : A) x = (_Float16)0.f;
        for (auto& x : B) x = (_Float16)0.f;
        for (int k = 0; k < np; ++k) { A[0 * 16 + k] = (_Float16)ldexpf(1.f, -13); B[k * 32 + 0] = (_Float16)ldexpf(1.f, -13); }
        for (auto& x : C) x = 1.0f;
        run(1);
        printf("C = 1 + %2d products of 2^-26 (exact %.4f ulp): D - 1 = %.2f ulp\n", np, np / 8.0, (D[0] - 1.0) / ldexp(1.0, -23));
    }
    for (int neg = 0; neg < 2; ++neg)
        for (int kk = 1; kk <= 7; ++kk) {        // one product of kk/4 ulp
            for (auto& x : A) x = (_Float16)0.f;
            for (auto& x : B) x = (_Float16)0.f;
            A[0] = (_Float16)(neg ? -ldexpf(1.f, -13) : ldexpf(1.f, -13));
            B[0] = (_Float16)ldexpf((float)kk, -12);          // product kk * 2^-25 = kk/4 ulp of 1.0
            for (auto& x : C) x = 1.0f;
            run(1);
            printf("C = 1 %c one product of %d/4 ulp(1): D - 1 = %.3f ulp(2^-23)\n", neg ? '-' : '+', kk, (D[0] - 1.0) / ldexp(1.0, -23));
        }
    // ---- chains ----
    for (int nchain : {4, 16, 64}) {
        Stats once, steps;
        for (int rep = 0; rep < 100; ++rep) {
            for (auto& x : A) x = (_Float16)(float)(urand() * 2 - 1);
            for (auto& x : B) x = (_Float16)(float)(urand() * 2 - 1);
            for (auto& x : C) x = 0.f;
            run(nchain);
            for (int i = 0; i < 32; ++i)
                for (int j = 0; j < 32; ++j) {
                    double ex = 0;
                    float st = 0.f;
                    for (int n = 0; n < nchain; ++n) {
                        ex = exact_step(n, i, j, ex);
                        st = (float)exact_step(n, i, j, (double)st);      // each step exactly rounded (RNE)
                    }
                    add(once, D[i * 32 + j], ex);
                    add(steps, D[i * 32 + j], (double)st);
                }
        }
        char nm[96];
        snprintf(nm, sizeof nm, "chain of %d MFMAs vs the exact sum", nchain); show(nm, once);
        snprintf(nm, sizeof nm, "chain of %d MFMAs vs a chain of RNE-rounded steps", nchain); show(nm, steps);
    }
    // positive-only data: a truncation bias shows as a signed mean
    {
        Stats once;
        for (int rep = 0; rep < 100; ++rep) {
            for (auto& x : A) x = (_Float16)(float)(urand());
            for (auto& x : B) x = (_Float16)(float)(urand());
            for (auto& x : C) x = 0.f;
            run(64);
            for (int i = 0; i < 32; ++i)
                for (int j = 0; j < 32; ++j) {
                    double ex = 0;
                    for (int n = 0; n < 64; ++n) ex = exact_step(n, i, j, ex);
                    add(once, D[i * 32 + j], ex);
                }
        }
        show("chain of 64 MFMAs, all operands positive, vs the exact sum", once);
    }
    return 0;
}
