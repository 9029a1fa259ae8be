// check + micro-benchmark of the Cooley-Tukey DFT pass (surfh_amd/csrc/dft_ct.hip): the four pass types of rfft2 / irfft2 on
// one length N against a float64 O(N^2) transform of every column, then timed at a cube-sized batch.
//   ct_main N [LP_check] [LP_time] [batch_time]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include "../../surfh_amd/csrc/dft_ct.h"
#define CK(x) do { hipError_t e_ = (hipError_t)(x); if (e_ != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void fill_k(float *p, long n, unsigned seed, int mode) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        float v = (float)(x & 0xFFFF) / 65536.f - 0.5f;
        if (mode == 1) v = ldexpf(v, -(int)((x >> 16) % 31));
        p[i] = v;
    }
}
// reference, float64: one thread per (output row, column).  kind 0: c2c (sgn), 1: r2c forward (real rows -> rows 0..N/2),
// 2: c2r inverse (rows 0..N/2 Hermitian -> real rows), 3: c2c inverse with the spectral mix applied to the source,
// 4 / 5: c2c of src * prod / src * conj(prod) (prod = mhat, same layout as src)
struct RefArgs {
    int kind, N; float sgn; double scale;
    const float *src; long ldb, sB; float *dst; long ldc, sC; int ncols, batch;    // ncols: complex (kind 0, 3) or real (1: source, 2: output) columns
    const float *mhat, *tpl; int T, LP; long PL, KBP;
    double add_w = 0;      // kind 6: src * conj(prod = mhat) + add_w (dk[n] + dkb[kb]) * add (= tpl), kb = col / LP, other axis length T
};
__global__ void ref_k(RefArgs a) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int Nh = a.N / 2;
    const int nout = a.kind == 1 ? Nh + 1 : a.N;
    const long total = (long)nout * a.ncols * a.batch;
    if (idx >= total) return;
    const int col = (int)(idx % a.ncols);
    const int bz = (int)((idx / a.ncols) % a.batch);
    const int k = (int)(idx / ((long)a.ncols * a.batch));
    double sr = 0, si = 0;
    for (int n = 0; n < a.N; ++n) {
        double xr, xi;
        if (a.kind == 1) {
            xr = a.src[(long)n * a.ldb + bz * a.sB + col]; xi = 0;
        } else if (a.kind == 2) {
            const int m = n <= Nh ? n : a.N - n;
            xr = a.src[(long)m * a.ldb + bz * a.sB + 2 * col];
            xi = a.src[(long)m * a.ldb + bz * a.sB + 2 * col + 1];
            if (n > Nh) xi = -xi;
            if (m == 0 || 2 * m == a.N) xi = 0;
        } else {
            xr = a.src[(long)n * a.ldb + bz * a.sB + 2 * col];
            xi = a.src[(long)n * a.ldb + bz * a.sB + 2 * col + 1];
            if (a.kind == 3) {
                const int kb = col / a.LP, l = col % a.LP;
                double mr = 0, mi = 0;
                for (int t = 0; t < a.T; ++t) {
                    mr += (double)a.tpl[(long)t * a.LP + l] * a.mhat[((long)t * 2 + 0) * a.PL + (long)n * a.KBP + kb];
                    mi += (double)a.tpl[(long)t * a.LP + l] * a.mhat[((long)t * 2 + 1) * a.PL + (long)n * a.KBP + kb];
                }
                const double r2 = xr * mr - xi * mi, i2 = xr * mi + xi * mr;
                xr = r2; xi = i2;
            }
            if (a.kind >= 4) {
                const double hr = a.mhat[(long)n * a.ldb + bz * a.sB + 2 * col];
                const double hi = (a.kind >= 5 ? -1.0 : 1.0) * a.mhat[(long)n * a.ldb + bz * a.sB + 2 * col + 1];
                const double r2 = xr * hr - xi * hi, i2 = xr * hi + xi * hr;
                xr = r2; xi = i2;
                if (a.kind == 6) {
                    const int kb = col / a.LP;
                    const double w = a.add_w * ((2.0 - 2.0 * cos(2.0 * M_PI * n / a.N)) + (2.0 - 2.0 * cos(2.0 * M_PI * kb / a.T)));
                    xr += w * a.tpl[(long)n * a.ldb + bz * a.sB + 2 * col];
                    xi += w * a.tpl[(long)n * a.ldb + bz * a.sB + 2 * col + 1];
                }
            }
        }
        const double th = a.sgn * 2.0 * M_PI * (double)(((long)n * k) % a.N) / (double)a.N;
        const double c = cos(th), s = sin(th);
        sr += xr * c - xi * s;
        si += xr * s + xi * c;
    }
    if (a.kind == 2) a.dst[(long)k * a.ldc + bz * a.sC + col] = (float)(sr * a.scale);
    else {
        a.dst[(long)k * a.ldc + bz * a.sC + 2 * col] = (float)(sr * a.scale);
        a.dst[(long)k * a.ldc + bz * a.sC + 2 * col + 1] = (float)(si * a.scale);
    }
}
__global__ void diff_k(const float *a, const float *b, long rows, long ld, long width, double *acc) {   // over rows x width
    double s0 = 0, s1 = 0, m0 = 0;
    const long n = rows * width;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long r = i / width, c = i % width;
        const double x = a[r * ld + c], y = b[r * ld + c], d = x - y;
        s0 += d * d; s1 += y * y; m0 = fmax(m0, fabs(d));
    }
    atomicAdd(&acc[0], s0); atomicAdd(&acc[1], s1);
    atomicMax((unsigned long long *)&acc[2], (unsigned long long)__double_as_longlong(m0));
}
static int compare(const char *what, const float *a, const float *b, long rows, long ld, long width, double *acc) {
    CK(hipMemset(acc, 0, 32));
    hipLaunchKernelGGL(diff_k, dim3(1024), dim3(256), 0, 0, a, b, rows, ld, width, acc);
    double h[4];
    CK(hipMemcpy(h, acc, 32, hipMemcpyDeviceToHost));
    const double rel = std::sqrt(h[0] / (h[1] > 0 ? h[1] : 1));
    printf("  %-28s rel L2 %.3e   max |d| %.3e   |ref| %.3e  %s\n", what, rel, h[2], std::sqrt(h[1]), rel < 2e-6 ? "ok" : "FAIL");
    return rel < 2e-6 ? 0 : 1;
}

int launch_dft_dif(hipStream_t stream, const DftCtArgs &g, const DftCtPlan &pl);      // dft_dif.hip: the c2c pass as a DIF step, independent waves

int main(int argc, char **argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 501;
    const int LPc = argc > 2 ? atoi(argv[2]) : 256;
    const int LPt = argc > 3 ? atoi(argv[3]) : 1024;
    const int Bt = argc > 4 ? atoi(argv[4]) : 128;
    int R, M;
    if (!dft_ct_factor(N, &R, &M)) { printf("N = %d: no factorisation\n", N); return 1; }
    DftCtPlan pl;
    CK(dft_ct_plan_create(N, &pl));
    printf("N = %d = %d x %d, MT %d KT %d\n", N, R, M, pl.MT, pl.KT);
    const int Nh = N / 2, NP = (N + 63) / 64 * 64;
    const double sc = 1.0 / std::sqrt((double)N);
    double *acc;
    CK(hipMalloc(&acc, 32));
    int bad = 0;
    const bool check = !getenv("CT_NOCHECK");
    for (int mode = 0; mode < (check ? 2 : 0); ++mode) {
        // ---- correctness: batch B entries of LPc columns, arrays [row][B][cols] ------------------------------------
        const int B = 3;
        const long ncx = LPc;                               // complex columns per batch entry
        const long ldx = (long)B * ncx * 2;                 // floats per row of a complex array
        float *a, *b, *c;
        const size_t nel = (size_t)(NP + 8) * ldx * 2;      // room for real arrays with 2 * ncx columns too
        CK(hipMalloc(&a, nel * 4)); CK(hipMalloc(&b, nel * 4)); CK(hipMalloc(&c, nel * 4));
        hipLaunchKernelGGL(fill_k, dim3(1024), dim3(256), 0, 0, a, (long)nel, 12345u + mode, mode);
        printf("data mode %d\n", mode);
        for (int dir = 0; dir < 2; ++dir) {                 // c2c forward / inverse
            DftCtArgs g;
            g.R = R; g.M = M; g.loader = DFT_CT_PLAIN; g.epi = DFT_CT_STORE; g.sgn = dir ? 1.f : -1.f; g.scale = (float)sc;
            g.src = a; g.ldb = ldx; g.sB = ncx * 2; g.dst = b; g.ldc = ldx; g.sC = ncx * 2; g.ncols = (int)ncx; g.batch = B;
            CK(hipMemset(b, 0, nel * 4)); CK(hipMemset(c, 0, nel * 4));
            CK(launch_dft_ct(0, g, pl));
            RefArgs r{0, N, g.sgn, sc, a, ldx, ncx * 2, c, ldx, ncx * 2, (int)ncx, B, nullptr, nullptr, 0, 0, 0, 0};
            hipLaunchKernelGGL(ref_k, dim3((unsigned)(((long)N * ncx * B + 255) / 256)), dim3(256), 0, 0, r);
            CK(hipDeviceSynchronize());
            bad += compare(dir ? "c2c inverse" : "c2c forward", b, c, N, ldx, ldx, acc);
            CK(hipMemset(b, 0, nel * 4));
            CK(launch_dft_dif(0, g, pl));
            CK(hipDeviceSynchronize());
            bad += compare(dir ? "c2c inverse (DIF)" : "c2c forward (DIF)", b, c, N, ldx, ldx, acc);
        }
        for (int cj = 0; cj < 2; ++cj) {     // c2c inverse of src * prod and src * conj(prod): prod = array c shifted (same layout)
            float *pr;
            CK(hipMalloc(&pr, nel * 4));
            hipLaunchKernelGGL(fill_k, dim3(1024), dim3(256), 0, 0, pr, (long)nel, 4242u + mode, mode);
            DftCtArgs g;
            g.R = R; g.M = M; g.loader = DFT_CT_PROD; g.epi = DFT_CT_STORE; g.sgn = 1.f; g.scale = (float)sc;
            g.src = a; g.ldb = ldx; g.sB = ncx * 2; g.dst = b; g.ldc = ldx; g.sC = ncx * 2; g.ncols = (int)ncx; g.batch = B;
            g.prod = pr; g.ldp = ldx; g.sP = ncx * 2; g.prod_sign = cj ? -1.f : 1.f;
            CK(hipMemset(b, 0, nel * 4)); CK(hipMemset(c, 0, nel * 4));
            CK(launch_dft_ct(0, g, pl));
            RefArgs r{4 + cj, N, 1.f, sc, a, ldx, ncx * 2, c, ldx, ncx * 2, (int)ncx, B, pr, nullptr, 0, 0, 0, 0};
            hipLaunchKernelGGL(ref_k, dim3((unsigned)(((long)N * ncx * B + 255) / 256)), dim3(256), 0, 0, r);
            CK(hipDeviceSynchronize());
            bad += compare(cj ? "c2c inverse of src * conj(prod)" : "c2c inverse of src * prod", b, c, N, ldx, ldx, acc);
            hipFree(pr);
        }
        {   // c2c inverse of src * conj(prod) + weights * add: columns n = kb * LP + l, one batch entry (3 k_beta of ncx wavelengths)
            const int hb = 3;
            const long LP = ncx, ld = (long)hb * LP * 2;
            float *pr, *ad;
            CK(hipMalloc(&pr, nel * 4)); CK(hipMalloc(&ad, nel * 4));
            hipLaunchKernelGGL(fill_k, dim3(1024), dim3(256), 0, 0, pr, (long)nel, 4243u + mode, mode);
            hipLaunchKernelGGL(fill_k, dim3(1024), dim3(256), 0, 0, ad, (long)nel, 4244u + mode, mode);
            DftCtArgs g;
            g.R = R; g.M = M; g.loader = DFT_CT_PRODADD; g.epi = DFT_CT_STORE; g.sgn = 1.f; g.scale = (float)sc;
            g.src = a; g.ldb = ld; g.sB = 0; g.dst = b; g.ldc = ld; g.sC = 0; g.ncols = (int)(hb * LP); g.batch = 1;
            g.prod = pr; g.ldp = ld; g.sP = 0; g.prod_sign = -1.f; g.add = ad; g.add_w = 0.37f; g.add_Nb = 2 * (hb - 1) + 1; g.LP = (int)LP;
            CK(hipMemset(b, 0, nel * 4)); CK(hipMemset(c, 0, nel * 4));
            CK(launch_dft_ct(0, g, pl));
            RefArgs r{6, N, 1.f, sc, a, ld, 0, c, ld, 0, (int)(hb * LP), 1, pr, ad, g.add_Nb, (int)LP, 0, 0, 0.37};
            hipLaunchKernelGGL(ref_k, dim3((unsigned)(((long)N * hb * LP + 255) / 256)), dim3(256), 0, 0, r);
            CK(hipDeviceSynchronize());
            bad += compare("c2c inv. of src * conj(prod) + w add", b, c, N, ld, ld, acc);
            hipFree(pr); hipFree(ad);
        }
        {   // r2c: real [N][B][2 ncx] -> complex rows 0..Nh [B][2 ncx][2]
            const long ldr = (long)B * ncx * 2, ldo = (long)B * ncx * 4;
            DftCtArgs g;
            g.R = R; g.M = M; g.loader = DFT_CT_PLAIN; g.epi = DFT_CT_HSEP; g.sgn = -1.f; g.scale = (float)(0.5 * sc);
            g.src = a; g.ldb = ldr; g.sB = ncx * 2; g.dst = b; g.ldc = ldo; g.sC = ncx * 4; g.ncols = (int)ncx; g.batch = B;
            CK(hipMemset(b, 0, nel * 4)); CK(hipMemset(c, 0, nel * 4));
            CK(launch_dft_ct(0, g, pl));
            RefArgs r{1, N, -1.f, sc, a, ldr, ncx * 2, c, ldo, ncx * 4, (int)(2 * ncx), B, nullptr, nullptr, 0, 0, 0, 0};
            hipLaunchKernelGGL(ref_k, dim3((unsigned)(((long)(Nh + 1) * 2 * ncx * B + 255) / 256)), dim3(256), 0, 0, r);
            CK(hipDeviceSynchronize());
            bad += compare("r2c (packed pairs, HSEP)", b, c, Nh + 1, ldo, ldo, acc);
        }
        {   // c2r: complex rows 0..Nh [B][2 ncx][2] -> real [N][B][2 ncx]
            const long ldi = (long)B * ncx * 4, ldo = (long)B * ncx * 2;
            DftCtArgs g;
            g.R = R; g.M = M; g.loader = DFT_CT_HPACK; g.epi = DFT_CT_STORE; g.sgn = 1.f; g.scale = (float)sc;
            g.src = a; g.ldb = ldi; g.sB = ncx * 4; g.dst = b; g.ldc = ldo; g.sC = ncx * 2; g.ncols = (int)ncx; g.batch = B;
            CK(hipMemset(b, 0, nel * 4)); CK(hipMemset(c, 0, nel * 4));
            CK(launch_dft_ct(0, g, pl));
            RefArgs r{2, N, 1.f, sc, a, ldi, ncx * 4, c, ldo, ncx * 2, (int)(2 * ncx), B, nullptr, nullptr, 0, 0, 0, 0};
            hipLaunchKernelGGL(ref_k, dim3((unsigned)(((long)N * 2 * ncx * B + 255) / 256)), dim3(256), 0, 0, r);
            CK(hipDeviceSynchronize());
            bad += compare("c2r (HPACK)", b, c, N, ldo, ldo, acc);
        }
        {   // inverse c2c with the spectral mix: columns n = kb * LP + l, one batch entry
            const int hb = 3, T = 3;
            const long KBP = 64, PL = (long)NP * KBP, LP = ncx;
            float *mhat, *tpl;
            CK(hipMalloc(&mhat, (size_t)T * 2 * PL * 4)); CK(hipMalloc(&tpl, (size_t)T * LP * 4));
            hipLaunchKernelGGL(fill_k, dim3(256), dim3(256), 0, 0, mhat, (long)T * 2 * PL, 777u, 0);
            hipLaunchKernelGGL(fill_k, dim3(256), dim3(256), 0, 0, tpl, (long)T * LP, 778u, 0);
            const long ld = (long)hb * LP * 2;
            DftCtArgs g;
            g.R = R; g.M = M; g.loader = DFT_CT_MIX; g.epi = DFT_CT_STORE; g.sgn = 1.f; g.scale = (float)sc;
            g.src = a; g.ldb = ld; g.sB = 0; g.dst = b; g.ldc = ld; g.sC = 0; g.ncols = (int)(hb * LP); g.batch = 1;
            g.mhat = mhat; g.tpl = tpl; g.T = T; g.LP = (int)LP; g.PL = PL; g.KBP = KBP;
            CK(hipMemset(b, 0, nel * 4)); CK(hipMemset(c, 0, nel * 4));
            CK(launch_dft_ct(0, g, pl));
            RefArgs r{3, N, 1.f, sc, a, ld, 0, c, ld, 0, (int)(hb * LP), 1, mhat, tpl, T, (int)LP, PL, KBP};
            hipLaunchKernelGGL(ref_k, dim3((unsigned)(((long)N * hb * LP + 255) / 256)), dim3(256), 0, 0, r);
            CK(hipDeviceSynchronize());
            bad += compare("c2c inverse + spectral mix", b, c, N, ld, ld, acc);
            hipFree(mhat); hipFree(tpl);
        }
        hipFree(a); hipFree(b); hipFree(c);
    }
    if (bad) { printf("FAILED: %d\n", bad); return 2; }
    // ---- timing: Bt batch entries x LPt columns (complex array = N * Bt * LPt * 8 bytes) -------------------------------
    {
        const long ncx = LPt, ldx = (long)Bt * ncx * 2;
        const size_t nel = (size_t)(NP + 8) * ldx;
        float *a, *b;
        CK(hipMalloc(&a, nel * 4)); CK(hipMalloc(&b, nel * 4));
        hipLaunchKernelGGL(fill_k, dim3(4096), dim3(256), 0, 0, a, (long)nel, 99u, 0);
        CK(hipMemset(b, 0, nel * 4));
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        const double gbc = (double)N * Bt * ncx * 8 / 1e9;       // one complex array
        struct Case { const char *name; int loader, epi; float sgn; double gb; } cases[] = {
            {"c2c forward (PLAIN/STORE)", DFT_CT_PLAIN, DFT_CT_STORE, -1.f, 2 * gbc},
            {"r2c (PLAIN/HSEP)", DFT_CT_PLAIN, DFT_CT_HSEP, -1.f, 1.5 * gbc},
            {"c2r (HPACK/STORE)", DFT_CT_HPACK, DFT_CT_STORE, 1.f, 1.5 * gbc},
            {"c2c of src * prod (PROD/STORE)", DFT_CT_PROD, DFT_CT_STORE, 1.f, 3 * gbc},
            {"... + w add (PRODADD/STORE)", DFT_CT_PRODADD, DFT_CT_STORE, 1.f, 4 * gbc},
        };
        float *pr;
        CK(hipMalloc(&pr, nel * 4));
        hipLaunchKernelGGL(fill_k, dim3(4096), dim3(256), 0, 0, pr, (long)nel, 77u, 0);
        for (auto &cs : cases) {
            DftCtArgs g;
            g.R = R; g.M = M; g.loader = cs.loader; g.epi = cs.epi; g.sgn = cs.sgn; g.scale = (float)sc;
            g.src = a; g.dst = b; g.ncols = (int)ncx; g.batch = Bt;
            // half-spectrum arrays hold N/2+1 rows of 2x wide rows: same bytes per row pair
            g.ldb = cs.loader == DFT_CT_HPACK ? 2 * ldx : ldx; g.sB = cs.loader == DFT_CT_HPACK ? ncx * 4 : ncx * 2;
            g.ldc = cs.epi == DFT_CT_HSEP ? 2 * ldx : ldx; g.sC = cs.epi == DFT_CT_HSEP ? ncx * 4 : ncx * 2;
            g.prod = pr; g.ldp = ldx; g.sP = ncx * 2; g.add = pr; g.add_w = 0.1f; g.add_Nb = 2 * (Bt - 1); g.LP = (int)ncx;
            if (&cs == &cases[0]) {       // the same pass as a DIF step
                for (int w = 0; w < 2; ++w) CK(launch_dft_dif(0, g, pl));
                hipEventRecord(e0, 0);
                const int repsd = getenv("CT_REPS") ? atoi(getenv("CT_REPS")) : 5;
                for (int w = 0; w < repsd; ++w) CK(launch_dft_dif(0, g, pl));
                hipEventRecord(e1, 0);
                CK(hipDeviceSynchronize());
                float msd = 0;
                hipEventElapsedTime(&msd, e0, e1);
                msd /= repsd;
                printf("time %-28s %.3f ms   %.2f GB -> %.2f TB/s\n", "c2c forward (DIF)", msd, cs.gb, cs.gb / msd);
            }
            for (int w = 0; w < 2; ++w) CK(launch_dft_ct(0, g, pl));
            hipEventRecord(e0, 0);
            const int reps = getenv("CT_REPS") ? atoi(getenv("CT_REPS")) : 5;      // CT_REPS=300: sustained load (the clock settles)
            for (int w = 0; w < reps; ++w) CK(launch_dft_ct(0, g, pl));
            hipEventRecord(e1, 0);
            CK(hipDeviceSynchronize());
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            ms /= reps;
            printf("time %-28s %.3f ms   %.2f GB -> %.2f TB/s\n", cs.name, ms, cs.gb, cs.gb / ms);
        }
        hipFree(a); hipFree(b); hipFree(pr);
    }
    dft_ct_plan_destroy(&pl);
    return 0;
}
