// BASELINE.json's north star allows MFMA "only for the dense batched 4x4 / 8x8 precision contractions where it actually fills a
// tile".  The one f64 matrix instruction of gfx950 that fits a 4 x 4 block is v_mfma_f64_4x4x4_4b_f64: FOUR independent 4x4x4
// products per wave instruction, one matrix element per lane (16 lanes per block).  The contraction of the hot path that has
// that shape is the Schur complement of a dynamic-factor message (factor/mod.rs:391-401, marginalise_factor_distance.rs:114-115):
//     T = Lab . W            (W = Lbb^-1, 4 x 4 by cofactors — not matrix-shaped, stays on the VALU either way)
//     Lout = Laa - T . Lba,  eout = ea - T . eb
// This micro-benchmark prices exactly that step, per wave of 64 messages, two ways:
//   valu  one LANE per message, the operands in registers (how k_robot_sweep computes it): 2 x 64 multiply-adds per lane
//   mfma  the same 64 messages as 16 rounds of FOUR blocks: W comes out of the cofactor code one message per lane, so it goes
//         through LDS to reach the one-element-per-lane layout, Lab / Lba are read from LDS in that layout directly (as they
//         could be from the kernel's component-major image), and the result goes back through LDS to one message per lane for
//         the subtraction.  The matrix instruction fuses multiply and add: tolerance build (libmgx_fma.so) only, never the product.
// The lane layout of the instruction is found by probing it (one-hot operands), not assumed.
// hipcc --offload-arch=gfx950 -O3 mfma_schur.hip -o mfma_schur && ./mfma_schur
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <vector>

// ---- layout probe: out[lane] for a = onehot(la), b = onehot(lb); and a check of the layout derived from it ----------------
__global__ void probe(int la, int lb, double *out) {
    const int l = threadIdx.x;
    const double a = l == la ? 1.0 : 0.0, b = l == lb ? 1.0 : 0.0;
    out[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
}
struct Maps {  // lane -> (block, row, col) of the element it holds
    int a_i[64], a_k[64], b_k[64], b_j[64], d_i[64], d_j[64], a_blk[64], b_blk[64], d_blk[64];  // (a lane may serve different blocks per role)
};
__constant__ Maps c_maps;

__global__ void check_layout(const double *A, const double *B, double *D) {  // A, B, D: [4 blocks][4][4] row-major
    const int l = threadIdx.x;
    const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(A[c_maps.a_blk[l] * 16 + 4 * c_maps.a_i[l] + c_maps.a_k[l]],
                                                        B[c_maps.b_blk[l] * 16 + 4 * c_maps.b_k[l] + c_maps.b_j[l]], 0.0, 0, 0, 0);
    D[c_maps.d_blk[l] * 16 + 4 * c_maps.d_i[l] + c_maps.d_j[l]] = d;
}

constexpr int REPS = 2000;

// one lane per message: T = Lab W ; Lo = Laa - T Lba ; eo = ea - T eb   (k-ascending dot products, as gbp_math.h)
__global__ void __launch_bounds__(64) schur_valu(const double *in, double *out, long long *cycles) {
    const int l = threadIdx.x;
    double lab[16], w[16], lba[16], laa[16], ea[4], eb[4];
    const double *p = in + (size_t)(blockIdx.x * 64 + l) * 72;
    for (int c = 0; c < 16; c++) { lab[c] = p[c]; w[c] = p[16 + c]; lba[c] = p[32 + c]; laa[c] = p[48 + c]; }
    for (int c = 0; c < 4; c++) { ea[c] = p[64 + c]; eb[c] = p[68 + c]; }
    double acc = 0.0;
    const long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < REPS; r++) {
        double t[16];
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                double s = lab[4 * i] * w[j];
#pragma unroll
                for (int k = 1; k < 4; k++) s += lab[4 * i + k] * w[4 * k + j];
                t[4 * i + j] = s;
            }
#pragma unroll
        for (int i = 0; i < 4; i++) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                double s = t[4 * i] * lba[j];
#pragma unroll
                for (int k = 1; k < 4; k++) s += t[4 * i + k] * lba[4 * k + j];
                acc += laa[4 * i + j] - s;
            }
            double s = t[4 * i] * eb[0];
#pragma unroll
            for (int k = 1; k < 4; k++) s += t[4 * i + k] * eb[k];
            acc += ea[i] - s;
        }
        w[0] += 1e-9;  // keep the iterations dependent on each other
    }
    const long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + l] = acc;
    if (l == 0) cycles[blockIdx.x] = t1 - t0;
}

// the same 64 messages on the matrix pipe: per round r (16 of them) the wave's four blocks hold messages 4 r .. 4 r + 3
__global__ void __launch_bounds__(64) schur_mfma(const double *in, double *out, long long *cycles) {
    __shared__ double s_w[64 * 17], s_lab[64 * 17], s_lba[64 * 17], s_t[64 * 17], s_o[64 * 17];
    const int l = threadIdx.x;
    double w[16], laa[16], ea[4], eb[4];
    const double *p = in + (size_t)(blockIdx.x * 64 + l) * 72;
    for (int c = 0; c < 16; c++) { s_lab[l * 17 + c] = p[c]; w[c] = p[16 + c]; s_lba[l * 17 + c] = p[32 + c]; laa[c] = p[48 + c]; }
    for (int c = 0; c < 4; c++) { ea[c] = p[64 + c]; eb[c] = p[68 + c]; }
    __syncthreads();
    const int ab = c_maps.a_blk[l], bb = c_maps.b_blk[l], db = c_maps.d_blk[l];
    const int ai = c_maps.a_i[l], ak = c_maps.a_k[l], bk = c_maps.b_k[l], bj = c_maps.b_j[l], di = c_maps.d_i[l], dj = c_maps.d_j[l];
    double acc = 0.0;
    const long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < REPS; r++) {
        // W leaves the cofactor code one message per lane: through LDS into the one-element-per-lane layout
#pragma unroll
        for (int c = 0; c < 16; c++) s_w[l * 17 + c] = w[c];
        __builtin_amdgcn_wave_barrier();
#pragma unroll 4
        for (int q = 0; q < 16; q++) {
            // round q: block b works on message 4 q + b
            const double t = __builtin_amdgcn_mfma_f64_4x4x4f64(s_lab[(4 * q + ab) * 17 + 4 * ai + ak], s_w[(4 * q + bb) * 17 + 4 * bk + bj], 0.0, 0, 0, 0);
            s_t[(4 * q + db) * 17 + 4 * di + dj] = t;  // T is the A operand of the second product: D layout -> A layout through LDS
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll 4
        for (int q = 0; q < 16; q++) {
            const double o = __builtin_amdgcn_mfma_f64_4x4x4f64(s_t[(4 * q + ab) * 17 + 4 * ai + ak], s_lba[(4 * q + bb) * 17 + 4 * bk + bj], 0.0, 0, 0, 0);
            s_o[(4 * q + db) * 17 + 4 * di + dj] = o;
        }
        __builtin_amdgcn_wave_barrier();
        // back to one message per lane for the subtractions (and T . eb, 16 multiply-adds, on the VALU)
#pragma unroll
        for (int i = 0; i < 4; i++) {
#pragma unroll
            for (int j = 0; j < 4; j++) acc += laa[4 * i + j] - s_o[l * 17 + 4 * i + j];
            double s = s_t[l * 17 + 4 * i] * eb[0];
#pragma unroll
            for (int k = 1; k < 4; k++) s += s_t[l * 17 + 4 * i + k] * eb[k];
            acc += ea[i] - s;
        }
        w[0] += 1e-9;
        __builtin_amdgcn_wave_barrier();
    }
    const long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + l] = acc;
    if (l == 0) cycles[blockIdx.x] = t1 - t0;
}

// the matrix products alone, operands already in the instruction's layout in registers: what the pipe itself costs
__global__ void __launch_bounds__(64) mfma_bare(double *out, long long *cycles) {
    const int l = threadIdx.x;
    double a = 1.0 + l * 1e-3, b = 0.5 - l * 1e-3, acc = 0.0;
    const long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < REPS; r++) {
#pragma unroll
        for (int q = 0; q < 32; q++) acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc, 0, 0, 0);
        a += 1e-9;
    }
    const long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + l] = acc;
    if (l == 0) cycles[blockIdx.x] = t1 - t0;
}

int main() {
    // the instruction's lane layout, probed: for every pair of lanes (la holds the only non-zero of A, lb of B) the lane the
    // product lands in.  la and lb interact iff they sit in the same block and name the same k; the four lanes la reaches are
    // row i(la) of D, the four lanes reached through lb are column j(lb).
    Maps m{};
    double *d_pr;
    hipMalloc(&d_pr, 64 * 8);
    std::vector<double> h(64);
    int hit[64][64];
    for (int la = 0; la < 64; la++)
        for (int lb = 0; lb < 64; lb++) {
            hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, la, lb, d_pr);
            hipMemcpy(h.data(), d_pr, 64 * 8, hipMemcpyDeviceToHost);
            hit[la][lb] = -1;
            for (int l = 0; l < 64; l++)
                if (h[l] != 0.0) hit[la][lb] = l;
        }
    auto key_of = [](const std::vector<int> &v) { unsigned long long k = 0; for (int x : v) k |= 1ull << x; return k; };
    unsigned long long Pa[64], Oa[64], Cb[64], Pb[64];  // as lane sets
    for (int la = 0; la < 64; la++) {
        std::vector<int> p, o;
        for (int lb = 0; lb < 64; lb++)
            if (hit[la][lb] >= 0) { p.push_back(lb); o.push_back(hit[la][lb]); }
        Pa[la] = key_of(p); Oa[la] = key_of(o);
    }
    for (int lb = 0; lb < 64; lb++) {
        std::vector<int> p, c;
        for (int la = 0; la < 64; la++)
            if (hit[la][lb] >= 0) { p.push_back(la); c.push_back(hit[la][lb]); }
        Pb[lb] = key_of(p); Cb[lb] = key_of(c);
    }
    // a block = the lanes that can reach each other: D lanes of the rows of the A lanes that share partners, transitively;
    // here simply: the union of the D rows of all A lanes whose partner sets intersect ... found by growing from each D lane
    auto rank_in = [](unsigned long long cls, const std::vector<unsigned long long> &all) {  // index of a class among the distinct ones, by lowest lane
        std::vector<unsigned long long> d;
        for (unsigned long long c : all) {
            bool seen = false;
            for (unsigned long long e : d) seen = seen || e == c;
            if (!seen) d.push_back(c);
        }
        int r = 0;
        for (unsigned long long e : d)
            if (__builtin_ctzll(e) < __builtin_ctzll(cls)) r++;
        return r;
    };
    // block of a lane, as the set of D lanes reachable from it (A lane -> its row's block = union of rows with intersecting partners)
    unsigned long long blockD[64];
    {
        int comp[64];  // A lanes of one block: connected through a shared row of D (same i) or a shared partner set (same k)
        for (int l = 0; l < 64; l++) comp[l] = l;
        for (bool changed = true; changed;) {
            changed = false;
            for (int x = 0; x < 64; x++)
                for (int y = 0; y < 64; y++)
                    if ((Oa[x] == Oa[y] || Pa[x] == Pa[y]) && comp[x] != comp[y]) {
                        const int c = comp[x] < comp[y] ? comp[x] : comp[y];
                        comp[x] = comp[y] = c;
                        changed = true;
                    }
        }
        for (int l = 0; l < 64; l++) {
            blockD[l] = 0ull;
            for (int x = 0; x < 64; x++)
                if (comp[x] == comp[l]) blockD[l] |= Oa[x];
        }
    }
    std::vector<unsigned long long> blocks(blockD, blockD + 64);
    bool ok = true;
    for (int l = 0; l < 64; l++) {
        ok = ok && __builtin_popcountll(Pa[l]) == 4 && __builtin_popcountll(Oa[l]) == 4 && __builtin_popcountll(blockD[l]) == 16;
        // A lane l: block, row class (by Oa), k class (by Pa) — classes ranked among those of ITS block
        std::vector<unsigned long long> rows, ks, cols, kbs;
        for (int x = 0; x < 64; x++)
            if (blockD[x] == blockD[l]) { rows.push_back(Oa[x]); ks.push_back(Pa[x]); }
        m.a_i[l] = rank_in(Oa[l], rows);
        m.a_k[l] = rank_in(Pa[l], ks);
    }
    for (int lb = 0; lb < 64; lb++) {
        // B lane: its block = the block of the D lanes it reaches; k class = the A lanes it meets (Pb), named like their Pa class
        const int some_a = __builtin_ctzll(Pb[lb]);
        m.b_k[lb] = m.a_k[some_a];
        std::vector<unsigned long long> cols;
        for (int y = 0; y < 64; y++)
            if (Cb[y] & blockD[some_a]) cols.push_back(Cb[y]);
        m.b_j[lb] = rank_in(Cb[lb], cols);
    }
    for (int d = 0; d < 64; d++) {
        // D lane: row = the Oa class containing it, column = the Cb class containing it; block index by lowest lane of the block
        int la = -1, lb = -1;
        for (int x = 0; x < 64 && la < 0; x++)
            if ((Oa[x] >> d) & 1ull) la = x;
        for (int y = 0; y < 64 && lb < 0; y++)
            if ((Cb[y] >> d) & 1ull) lb = y;
        if (la < 0 || lb < 0) { ok = false; break; }
        m.d_i[d] = m.a_i[la];
        m.d_j[d] = m.b_j[lb];
    }
    // the block NUMBER of a lane in each role: A and B lanes by the D block they feed, D lanes by membership
    int a_blk[64], b_blk[64], d_blk[64];
    for (int l = 0; l < 64; l++) {
        a_blk[l] = rank_in(blockD[l], blocks);
        d_blk[l] = -1;
    }
    for (int l = 0; l < 64; l++)
        for (int d = 0; d < 64; d++)
            if ((blockD[l] >> d) & 1ull) d_blk[d] = a_blk[l];
    for (int lb = 0; lb < 64; lb++) b_blk[lb] = a_blk[__builtin_ctzll(Pb[lb])];
    if (!ok) {
        printf("the probe found no 4 x (4x4x4) structure: lane 0 reaches %d B lanes and %d D lanes, its block has %d D lanes\n",
               __builtin_popcountll(Pa[0]), __builtin_popcountll(Oa[0]), __builtin_popcountll(blockD[0]));
        return 1;
    }
    for (int l = 0; l < 64; l++) { m.a_blk[l] = a_blk[l]; m.b_blk[l] = b_blk[l]; m.d_blk[l] = d_blk[l]; }
    hipMemcpyToSymbol(HIP_SYMBOL(c_maps), &m, sizeof m);
    // check numerically
    {
        double hA[64], hB[64], hD[64], *dA, *dB, *dD;
        for (int q = 0; q < 64; q++) { hA[q] = std::sin(1.0 + q); hB[q] = std::cos(0.3 * q); }
        hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 512);
        hipMemcpy(dA, hA, 512, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 512, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(check_layout, dim3(1), dim3(64), 0, 0, dA, dB, dD);
        hipMemcpy(hD, dD, 512, hipMemcpyDeviceToHost);
        double worst = 0.0;
        for (int b = 0; b < 4; b++)
            for (int i = 0; i < 4; i++)
                for (int j = 0; j < 4; j++) {
                    double sum = 0.0;
                    for (int k = 0; k < 4; k++) sum += hA[b * 16 + 4 * i + k] * hB[b * 16 + 4 * k + j];
                    worst = std::fmax(worst, std::fabs(sum - hD[b * 16 + 4 * i + j]));
                }
        printf("v_mfma_f64_4x4x4_4b_f64 lane layout (probed), lanes 0..19: ");
        for (int l = 0; l < 20; l++)
            printf("%d: A b%d(%d,%d) B b%d(%d,%d) D b%d(%d,%d)  ", l, m.a_blk[l], m.a_i[l], m.a_k[l], m.b_blk[l], m.b_k[l], m.b_j[l], m.d_blk[l], m.d_i[l], m.d_j[l]);
        printf("\nfour products through that layout vs the host: max |diff| %.1e\n", worst);
        if (worst > 1e-12) return 1;
    }

    for (int wgs : {256, 1024, 2048}) {  // one wave per workgroup: 1, 4, 8 waves per CU = 0.25, 1, 2 per SIMD
        std::vector<double> in((size_t)wgs * 64 * 72);
        for (size_t i = 0; i < in.size(); i++) in[i] = std::sin(0.37 * (double)(i % 1009)) + ((i % 72) % 5 == 0 ? 3.0 : 0.0);
        double *d_in, *o1, *o2;
        long long *c1, *c2, *c3;
        hipMalloc(&d_in, in.size() * 8); hipMalloc(&o1, (size_t)wgs * 64 * 8); hipMalloc(&o2, (size_t)wgs * 64 * 8);
        hipMalloc(&c1, wgs * 8); hipMalloc(&c2, wgs * 8); hipMalloc(&c3, wgs * 8);
        hipMemcpy(d_in, in.data(), in.size() * 8, hipMemcpyHostToDevice);
        for (int rep = 0; rep < 2; rep++) {
            hipLaunchKernelGGL(schur_valu, dim3(wgs), dim3(64), 0, 0, d_in, o1, c1);
            hipLaunchKernelGGL(schur_mfma, dim3(wgs), dim3(64), 0, 0, d_in, o2, c2);
            hipLaunchKernelGGL(mfma_bare, dim3(wgs), dim3(64), 0, 0, o1 + 0 * 64, c3);
            hipDeviceSynchronize();
        }
        hipLaunchKernelGGL(schur_valu, dim3(wgs), dim3(64), 0, 0, d_in, o1, c1);
        hipDeviceSynchronize();
        std::vector<long long> a(wgs), b(wgs), c(wgs);
        std::vector<double> r1((size_t)wgs * 64), r2((size_t)wgs * 64);
        hipMemcpy(a.data(), c1, wgs * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), c2, wgs * 8, hipMemcpyDeviceToHost);
        hipMemcpy(c.data(), c3, wgs * 8, hipMemcpyDeviceToHost);
        hipMemcpy(r1.data(), o1, r1.size() * 8, hipMemcpyDeviceToHost); hipMemcpy(r2.data(), o2, r2.size() * 8, hipMemcpyDeviceToHost);
        double sa = 0, sb = 0, sc = 0, err = 0, mag = 0;
        for (int i = 0; i < wgs; i++) { sa += (double)a[i]; sb += (double)b[i]; sc += (double)c[i]; }
        for (size_t i = 0; i < r1.size(); i++) { err = std::fmax(err, std::fabs(r1[i] - r2[i])); mag = std::fmax(mag, std::fabs(r1[i])); }
        printf("%4d waves (%.2f per SIMD): per 64 messages  VALU lane-per-message %.0f cycles | MFMA through LDS %.0f cycles | "
               "32 bare MFMAs (the pipe alone) %.0f cycles | max |valu - mfma| / max|valu| = %.1e\n",
               wgs, wgs / 1024.0, sa / wgs / REPS, sb / wgs / REPS, sc / wgs / REPS, err / mag);
        hipFree(d_in); hipFree(o1); hipFree(o2); hipFree(c1); hipFree(c2); hipFree(c3);
    }
    return 0;
}
