// GPU-box microbenchmark (diagnostic, not part of the product): cycles per level of a
// dependent LDS pointer chase shaped like the forest walk, for several read mixes,
// waves per CU and chains per lane.  Build: hipcc --offload-arch=gfx950 -O3 -o lds_chain lds_chain.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef unsigned long long u64;
typedef __attribute__((address_space(3))) float lds_f32;
typedef __attribute__((address_space(3))) u64 lds_u64;
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4u lds_u4;
#define LDS_AT(type, a) (reinterpret_cast<type *>((__UINTPTR_TYPE__)(unsigned)(a)))

// LDS: [0, 32768) "features" (floats), [32768, 32768+65536) 8-byte node words {thr, next pair addr}
constexpr int FEA = 32768, NODES = 65536;

template <int MODE, int CHAINS>
__device__ __forceinline__ void step(uint2 (&cur)[CHAINS], const unsigned (&lanek)[CHAINS])
{
    float x[CHAINS];
    u64 lw[CHAINS], rw[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; c++) {
        if (MODE != 3) {
            const unsigned xa = __builtin_amdgcn_perm(cur[c].y, lanek[c], 0x0c020700u);
            x[c] = (MODE == 5) ? 0.5f : *LDS_AT(const lds_f32, xa & 0x7ffc);
        } else x[c] = __uint_as_float(cur[c].y);
        const unsigned ca = cur[c].y & 0x3fff8u;
        if (MODE == 0 || MODE == 5) {  // the walk: x + two b64
            lw[c] = *LDS_AT(const volatile lds_u64, ca);
            rw[c] = *LDS_AT(const volatile lds_u64, ca + 8);
        } else if (MODE == 1) {  // x + one b64
            lw[c] = *LDS_AT(const volatile lds_u64, ca);
            rw[c] = lw[c] ^ 0x100000000ull;
        } else if (MODE == 2) {  // x + one b128
            const v4u q = *LDS_AT(const lds_u4, ca & ~15u);
            lw[c] = ((u64)q.y << 32) | q.x;
            rw[c] = ((u64)q.w << 32) | q.z;
        } else {  // 3: no LDS at all
            lw[c] = ((u64)(cur[c].y * 2654435761u) << 32) | cur[c].x;
            rw[c] = ((u64)(cur[c].y + 12345u) << 32) | cur[c].x;
        }
    }
#pragma unroll
    for (int c = 0; c < CHAINS; c++) {
        const bool gl = x[c] <= __uint_as_float(cur[c].x);
        cur[c].x = gl ? (unsigned)lw[c] : (unsigned)rw[c];
        cur[c].y = gl ? (unsigned)(lw[c] >> 32) : (unsigned)(rw[c] >> 32);
    }
}

template <int MODE, int CHAINS>
__global__ void chase(const uint2 *init, int levels, int reps, long long *out, unsigned *sink)
{
    extern __shared__ char lds[];
    const int tid = threadIdx.x, nthr = blockDim.x;
    for (int i = tid; i < FEA / 4; i += nthr) *LDS_AT(lds_f32, 4 * i) = (float)((i * 2654435761u) >> 8) * (1.0f / 16777216.0f);
    for (int i = tid; i < NODES / 8; i += nthr) {
        const uint2 w = init[i];
        *LDS_AT(lds_u64, FEA + 8 * i) = ((u64)w.y << 32) | w.x;
    }
    __syncthreads();
    uint2 cur[CHAINS];
    unsigned lanek[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; c++) {
        cur[c] = init[(tid * 7 + c * 131) & (NODES / 8 - 1)];
        lanek[c] = ((tid & 63) << 2) | (c << 16);
    }
    long long t0 = 0, t1 = 0;
    for (int r = 0; r < reps; r++) {
        if (r == 1) t0 = __builtin_amdgcn_s_memtime();
        int d = levels;
        for (; d >= 4; d -= 4) {
            step<MODE, CHAINS>(cur, lanek);
            step<MODE, CHAINS>(cur, lanek);
            step<MODE, CHAINS>(cur, lanek);
            step<MODE, CHAINS>(cur, lanek);
        }
    }
    t1 = __builtin_amdgcn_s_memtime();
    unsigned s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; c++) s += cur[c].x + cur[c].y;
    if (s == 0x12345678u) sink[0] = s;
    if ((tid & 63) == 0) out[blockIdx.x * (nthr / 64) + (tid >> 6)] = t1 - t0;
}

template <int MODE, int CHAINS>
double run(const uint2 *d_init, int waves, int levels, long long *d_out, unsigned *d_sink)
{
    const int reps = 41, grid = 256;
    hipFuncSetAttribute(reinterpret_cast<const void *>(chase<MODE, CHAINS>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, FEA + NODES);
    hipLaunchKernelGGL((chase<MODE, CHAINS>), dim3(grid), dim3(64 * waves), FEA + NODES, 0, d_init, levels, reps, d_out, d_sink);
    hipDeviceSynchronize();
    std::vector<long long> h((size_t)grid * waves);
    hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : h) s += (double)v;
    return s / (double)h.size() / (double)((reps - 1) * levels);
}

int main(int argc, char **argv)
{
    const int coherent = argc > 1 ? atoi(argv[1]) : 0;
    std::vector<uint2> init(NODES / 8);
    srand(1);
    for (size_t i = 0; i < init.size(); i++) {
        const unsigned pair = coherent ? (unsigned)((i * 5 + 1) % (NODES / 8 - 2)) : (unsigned)(rand() % (NODES / 8 - 2));
        const float thr = (float)(rand() % 1000) / 1000.0f;
        unsigned tb;
        memcpy(&tb, &thr, 4);
        init[i] = make_uint2(tb, (unsigned)(FEA + 8 * pair) | ((unsigned)(rand() % 120) << 24));
    }
    if (coherent)  // every lane starts at (and stays on) the same nodes
        for (size_t i = 0; i < init.size(); i++) init[i].x = 0x7f800000u;
    uint2 *d_init;
    long long *d_out;
    unsigned *d_sink;
    hipMalloc(&d_init, init.size() * 8);
    hipMalloc(&d_out, 256 * 16 * 8);
    hipMalloc(&d_sink, 4);
    hipMemcpy(d_init, init.data(), init.size() * 8, hipMemcpyHostToDevice);
    printf("cycles per level (s_memtime ticks), %s addresses; rows: waves per CU\n", coherent ? "coherent" : "random");
    printf("waves  walk(x+2xb64)  x+1xb64  x+b128  noLDS  walk-noX  walk-2chains  walk-4chains  b128-2chains  b128-4chains\n");
    const int wl[] = {1, 2, 4, 7, 8, 12, 14, 16};
    for (int w : wl) {
        printf("%5d  %8.1f  %8.1f  %8.1f  %8.1f  %8.1f  %8.1f  %8.1f  %8.1f  %8.1f\n", w,
               run<0, 1>(d_init, w, 20, d_out, d_sink), run<1, 1>(d_init, w, 20, d_out, d_sink),
               run<2, 1>(d_init, w, 20, d_out, d_sink), run<3, 1>(d_init, w, 20, d_out, d_sink),
               run<5, 1>(d_init, w, 20, d_out, d_sink), run<0, 2>(d_init, w, 20, d_out, d_sink),
               run<0, 4>(d_init, w, 20, d_out, d_sink), run<2, 2>(d_init, w, 20, d_out, d_sink),
               run<2, 4>(d_init, w, 20, d_out, d_sink));
        fflush(stdout);
    }
    return 0;
}
