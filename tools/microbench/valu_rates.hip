// valu_rates.hip -- what one MI355X sustains for the instruction kinds the traversal
// kernels are made of.  Every row issues the NAMED instruction through inline assembly,
// so the compiler can neither pack two scalar operations into one v_pk_* nor fuse or
// drop anything; run_valu_rates.sh counts the mnemonics in the generated assembly and
// prints the counts beside the timings.  Rates are per wave64 instruction:
//   wave-inst/clk/CU = 1.0  <=>  one instruction per SIMD every four clocks.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

enum Kind {
    K_FMA, K_MUL, K_ADD, K_PK_FMA, K_PK_MUL, K_PK_ADD, K_PK_FMA_SGPR, K_MIN3, K_MAX, K_CMP, K_CNDMASK, K_READLANE,
    K_RCP, K_SQRT, K_MIX_PK_MUL_ADD, K_MIX_CMP_CNDMASK, K_MIX_FMA_READLANE, K_COUNT
};

// one "group" = 8 independent instructions of the kind; 16 groups per loop body
#define REP8(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)

template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b)
{
    float x[8];
    f2 p[8];
    uint64_t m[8];
    int s[8];
    for (int i = 0; i < 8; i++) {
        x[i] = threadIdx.x + i;
        p[i] = f2{x[i], x[i] + 0.5f};
        m[i] = 0x5555555555555555ull << (i & 1);
        s[i] = i;
    }
    f2 pa = {a, a}, pb = {b, b};
    // a and b as wave-uniform values in scalar registers (what a broadcast triangle operand is)
    float sa = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a)));
    uint64_t sa2 = (uint64_t)(uint32_t)__builtin_bit_cast(int, sa); // low half = a, broadcast by op_sel_hi
    float sb = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, b)));
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            if (KIND == K_FMA) {
#define M(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
                REP8(M)
#undef M
            } else if (KIND == K_MUL) {
#define M(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
                REP8(M)
#undef M
            } else if (KIND == K_ADD) {
#define M(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(b));
                REP8(M)
#undef M
            } else if (KIND == K_PK_FMA) {
#define M(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pa), "v"(pb));
                REP8(M)
#undef M
            } else if (KIND == K_PK_MUL) {
#define M(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pa));
                REP8(M)
#undef M
            } else if (KIND == K_PK_ADD) {
#define M(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
                REP8(M)
#undef M
            } else if (KIND == K_PK_FMA_SGPR) {
                // one scalar operand broadcast to both halves: the "two rays per lane against a
                // wave-uniform triangle" form
#define M(i) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(p[i]) : "s"(sa2));
                REP8(M)
#undef M
            } else if (KIND == K_MIN3) {
#define M(i) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(x[(i + 1) & 7]), "v"(a));
                REP8(M)
#undef M
            } else if (KIND == K_MAX) {
#define M(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
                REP8(M)
#undef M
            } else if (KIND == K_CMP) {
#define M(i) asm volatile("v_cmp_lt_f32 %0, %1, %2" : "=s"(m[i]) : "v"(x[i]), "v"(a));
                REP8(M)
#undef M
            } else if (KIND == K_CNDMASK) {
#define M(i) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "s"(m[i]));
                REP8(M)
#undef M
            } else if (KIND == K_READLANE) {
#define M(i) asm volatile("v_readlane_b32 %0, %1, 7" : "=s"(s[i]) : "v"(x[i]));
                REP8(M)
#undef M
            } else if (KIND == K_RCP) {
#define M(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[i]));
                REP8(M)
#undef M
            } else if (KIND == K_SQRT) {
#define M(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(x[i]));
                REP8(M)
#undef M
            } else if (KIND == K_MIX_PK_MUL_ADD) {
                // the unfused reference arithmetic on two rays per lane: multiply, then add
#define M(i)                                                                                                                       \
    if ((i) & 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));                                                   \
    else asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pa));
                REP8(M)
#undef M
            } else if (KIND == K_MIX_CMP_CNDMASK) {
#define M(i)                                                                                                                       \
    if ((i) & 1) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "s"(m[i - 1]));                                \
    else asm volatile("v_cmp_lt_f32 %0, %1, %2" : "=s"(m[i]) : "v"(x[i]), "v"(a));
                REP8(M)
#undef M
            } else if (KIND == K_MIX_FMA_READLANE) {
                // a broadcast-ray filter step: one v_readlane per three v_fma
#define M(i)                                                                                                                       \
    if (((i) & 3) == 3) asm volatile("v_readlane_b32 %0, %1, 7" : "=s"(s[i]) : "v"(x[i]));                                         \
    else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
                REP8(M)
#undef M
            }
        }
    }
    float r = 0.f;
    for (int i = 0; i < 8; i++) r += x[i] + p[i].x + p[i].y + (float)(m[i] & 0xff) + (float)s[i];
    out[blockIdx.x * 256 + threadIdx.x] = r + sb;
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
static void run(const char *name, int blocks_per_cu, double flop_per_lane_inst)
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
    double wave_insts = (double)blocks * 4 * iters * 16 * 8; // wave64 instructions issued
    double per_s = wave_insts / (ms * 1e-3);
    printf("%-22s waves/SIMD=%d  %8.3f ms  %8.2f Ginst(wave)/s  = %5.2f wave-inst/clk/CU @2.4GHz", name, blocks_per_cu, ms,
           per_s / 1e9, per_s / 256 / 2.4e9);
    if (flop_per_lane_inst > 0) printf("  %7.1f TFLOP/s", per_s * 64 * flop_per_lane_inst / 1e12);
    printf("\n");
    hipEventDestroy(a); hipEventDestroy(b);
    hipFree(out);
}

int main()
{
    for (int bpc : {1, 2, 4, 8}) {
        run<K_FMA>("v_fma_f32", bpc, 2);
        run<K_MUL>("v_mul_f32", bpc, 1);
        run<K_ADD>("v_add_f32", bpc, 1);
        run<K_PK_FMA>("v_pk_fma_f32", bpc, 4);
        run<K_PK_MUL>("v_pk_mul_f32", bpc, 2);
        run<K_PK_ADD>("v_pk_add_f32", bpc, 2);
        run<K_PK_FMA_SGPR>("v_pk_mul_f32(sgpr)", bpc, 2);
        run<K_MIN3>("v_min3_f32", bpc, 2);
        run<K_MAX>("v_max_f32", bpc, 1);
        run<K_CMP>("v_cmp_lt_f32", bpc, 0);
        run<K_CNDMASK>("v_cndmask_b32", bpc, 0);
        run<K_READLANE>("v_readlane_b32", bpc, 0);
        run<K_RCP>("v_rcp_f32", bpc, 1);
        run<K_SQRT>("v_sqrt_f32", bpc, 1);
        run<K_MIX_PK_MUL_ADD>("mix pk_mul:pk_add 1:1", bpc, 2);
        run<K_MIX_CMP_CNDMASK>("mix cmp:cndmask 1:1", bpc, 0);
        run<K_MIX_FMA_READLANE>("mix fma:readlane 3:1", bpc, 1.5);
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
