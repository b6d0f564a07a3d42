// mfma_filter_rate.hip -- what one MI355X sustains for a ray/triangle edge-function filter whose three
// edge functions per pair are evaluated as a [16 edges-rows x 32] x [32 x 16 rays] f16 MFMA
// (v_mfma_f32_16x16x32_f16; 4 triangles x 16 rays = 64 pairs per instruction) followed by the
// sign test on the VALU (min3, max3, max, compare).  Prints pairs/s for several register tilings.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NB, int FILLER>
__global__ __launch_bounds__(256) void k_mfma(float *out, int iters, const h8 *recs, float thr)
{
    __shared__ h8 tile[64 * 32];  // 64 A tiles (256 triangles) x 16 rows x {hi, lo} = 32 KB
    for (int i = threadIdx.x; i < 64 * 32; i += 256) tile[i] = recs[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    h8 B[NB];
    for (int b = 0; b < NB; b++)
        for (int j = 0; j < 8; j++) B[b][j] = (_Float16)(0.001f * ((lane * 7 + b * 3 + j) % 97) - 0.04f);
    const h8 *my = tile + (lane & 15) * 2 + (lane >> 5);
    float hits = 0.f;
    const f4 zero = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; it++) {
        for (int t = 0; t < 64; t++) {
            const h8 a = my[t * 32];
            float g = -1e30f;
#pragma unroll
            for (int b = 0; b < NB; b++) {
                f4 e = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, B[b], zero, 0, 0, 0);
                if (FILLER) {
                    float lo = __builtin_fminf(__builtin_fminf(e[0], e[1]), e[2]);
                    float hi = __builtin_fmaxf(__builtin_fmaxf(e[0], e[1]), e[2]);
                    g = __builtin_fmaxf(g, __builtin_fmaxf(lo, -hi));
                } else {
                    g = __builtin_fmaxf(g, e[0]);
                }
            }
            if (__builtin_amdgcn_ballot_w64(g >= thr)) hits += 1.f;  // never taken with thr = 1e30
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = hits;
}

template <int NB, int FILLER>
static void run(int blocks_per_cu, const h8 *recs)
{
    const int iters = 200, blocks = 256 * blocks_per_cu;
    float *out;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    k_mfma<NB, FILLER><<<blocks, 256>>>(out, 5, recs, 1e30f);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k_mfma<NB, FILLER><<<blocks, 256>>>(out, iters, recs, 1e30f);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double mfmas = double(blocks) * 4 * iters * 64 * NB;
    const double pairs = mfmas * 64;
    printf("mfma16x16x32_f16 NB=%2d %s waves/SIMD=%d  %8.3f ms  %7.1f Gpairs/s  %6.2f cyc/MFMA/SIMD @2.4GHz  (%.0f TFLOP/s f16)\n", NB,
           FILLER ? "min3+max3+max+cmp" : "max only         ", blocks_per_cu, ms, pairs / ms / 1e6,
           2.4e9 * 1024 * (ms * 1e-3) / mfmas, mfmas * 16384 / ms / 1e9);
    hipFree(out);
}

int main()
{
    std::vector<_Float16> h(64 * 32 * 8);
    for (size_t i = 0; i < h.size(); i++) h[i] = (_Float16)(0.01f * float((i * 37) % 199) - 1.0f);
    h8 *recs;
    hipMalloc(&recs, h.size() * 2);
    hipMemcpy(recs, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    for (int w = 1; w <= 2; w++) {
        run<4, 1>(w, recs);
        run<8, 1>(w, recs);
        run<16, 1>(w, recs);
        run<16, 0>(w, recs);
    }
    run<8, 1>(3, recs);
    run<16, 1>(3, recs);
    return 0;
}
