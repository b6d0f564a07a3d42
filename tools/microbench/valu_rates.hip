// valu_rates.hip -- measures what one MI355X sustains for the instruction kinds the
// traversal filter is made of: v_fma_f32, v_pk_fma_f32, v_min3_f32, v_max_f32, v_cmp, and
// LDS broadcast reads.  Prints achieved instructions/clk/CU and TFLOP/s per kind.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b)
{
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7};
    f2 pa = {a, a}, pb = {b, b};
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            if (KIND == 0) { // 8 independent v_fma_f32
                x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
                x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
            } else if (KIND == 1) { // 4 independent v_pk_fma_f32 (= 8 fma)
                p0 = __builtin_elementwise_fma(p0, pa, pb); p1 = __builtin_elementwise_fma(p1, pa, pb);
                p2 = __builtin_elementwise_fma(p2, pa, pb); p3 = __builtin_elementwise_fma(p3, pa, pb);
            } else if (KIND == 2) { // v_min3_f32
                x0 = __builtin_fminf(__builtin_fminf(x0, x1), a); x1 = __builtin_fminf(__builtin_fminf(x1, x2), a);
                x2 = __builtin_fminf(__builtin_fminf(x2, x3), a); x3 = __builtin_fminf(__builtin_fminf(x3, x4), a);
                x4 = __builtin_fminf(__builtin_fminf(x4, x5), a); x5 = __builtin_fminf(__builtin_fminf(x5, x6), a);
                x6 = __builtin_fminf(__builtin_fminf(x6, x7), a); x7 = __builtin_fminf(__builtin_fminf(x7, x0), a);
            } else if (KIND == 3) { // v_mul_f32 (non-fused)
                x0 = x0 * a; x1 = x1 * a; x2 = x2 * a; x3 = x3 * a; x4 = x4 * a; x5 = x5 * a; x6 = x6 * a; x7 = x7 * a;
            }
        }
    }
    if (KIND == 1) { x0 = p0.x + p0.y; x1 = p1.x + p1.y; x2 = p2.x + p2.y; x3 = p3.x + p3.y; }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

// the shared-origin filter body: 4 rays per lane against one LDS record per iteration
__global__ __launch_bounds__(256) void k_filter(float *out, int iters, const float *rec_in)
{
    __shared__ float tile[256 * 12];
    for (int i = threadIdx.x; i < 256 * 12; i += 256) tile[i] = rec_in[i];
    __syncthreads();
    float hx[4], hy[4], hz[4], acc = 0.f;
    for (int q = 0; q < 4; q++) { hx[q] = threadIdx.x * 0.01f + q; hy[q] = 1.f - q; hz[q] = 0.5f * q; }
    for (int i = 0; i < iters; i++) {
        for (int j = 0; j < 256; j++) {
            const float *r = tile + j * 12;
            float G = -1e30f;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                float F0 = __builtin_fmaf(r[2], hz[q], __builtin_fmaf(r[1], hy[q], r[0] * hx[q]));
                float F1 = __builtin_fmaf(r[5], hz[q], __builtin_fmaf(r[4], hy[q], r[3] * hx[q]));
                float F2 = __builtin_fmaf(r[8], hz[q], __builtin_fmaf(r[7], hy[q], r[6] * hx[q]));
                float lo = __builtin_fminf(__builtin_fminf(F0, F1), F2);
                float hi = __builtin_fmaxf(__builtin_fmaxf(F0, F1), F2);
                G = __builtin_fmaxf(G, __builtin_fmaxf(lo, -hi));
            }
            if (__builtin_amdgcn_ballot_w64(G >= 1e30f)) acc += 1.f; // never taken
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc + hx[0];
}

template <int KIND>
static void run(const char *name, int blocks_per_cu, double insts_per_iter_thread, double flop_per_inst)
{
    int iters = 2000;
    int blocks = 256 * blocks_per_cu;
    float *out;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k<KIND><<<blocks, 256>>>(out, 10, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<KIND><<<blocks, 256>>>(out, iters, 1.0001f, 0.5f);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    double wave_insts = (double)blocks * 4 * iters * 16 * insts_per_iter_thread; // per-wave instruction count
    double per_s = wave_insts / (ms * 1e-3);
    printf("%-14s waves/SIMD=%d  %8.3f ms  %7.2f Ginst(wave)/s  = %6.2f wave-inst/clk/CU @2.4GHz  %7.1f TFLOP/s\n", name,
           blocks_per_cu, ms, per_s / 1e9, per_s / 256 / 2.4e9, per_s * 64 * flop_per_inst / 1e12);
    hipFree(out);
}

int main()
{
    for (int bpc : {1, 2, 4, 8}) {
        run<0>("v_fma_f32", bpc, 8, 2);
        run<1>("v_pk_fma_f32", bpc, 4, 4);
        run<2>("v_min3_f32", bpc, 8, 2);
        run<3>("v_mul_f32", bpc, 8, 1);
    }
    // filter body
    std::vector<float> rec(256 * 12);
    for (size_t i = 0; i < rec.size(); i++) rec[i] = 0.001f * (float)(i % 97) - 0.03f;
    float *drec, *out;
    hipMalloc(&drec, rec.size() * 4);
    hipMemcpy(drec, rec.data(), rec.size() * 4, hipMemcpyHostToDevice);
    for (int bpc : {1, 2, 4, 8}) {
        int blocks = 256 * bpc, iters = 40;
        hipMalloc(&out, blocks * 256 * 4);
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        k_filter<<<blocks, 256>>>(out, 1, drec);
        hipDeviceSynchronize();
        hipEventRecord(a);
        k_filter<<<blocks, 256>>>(out, iters, drec);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        double pairs = (double)blocks * 256 * 4 * iters * 256;
        printf("filter(shared-origin,K=4) blocks/CU=%d  %8.3f ms  %8.2f Gpairs/s  (x52 FLOP = %6.1f TFLOP/s-equivalent)\n", bpc, ms,
               pairs / (ms * 1e-3) / 1e9, pairs * 52 / (ms * 1e-3) / 1e12);
        hipFree(out);
    }
    return 0;
}
