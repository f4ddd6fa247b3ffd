// GPU-box microbenchmark (diagnostic, not part of the product): what a TWO-LEVEL forest node would cost the
// walk of forest_q2_kernel (configs[4]: 529 features, the wide node word, one dependent chain per lane, 14
// walking waves per CU).  Today a level is one LDS round trip: {ds_read_u16 code, ds_read_b64 child pair} ->
// compare -> select.  A super-node that carries the features and ranks of a node AND of its two children
// resolves two levels per round trip: {3 x ds_read_u16 codes, the four grandchild super-nodes -- 4 x
// ds_read_b128 (16-byte nodes) or 4 x ds_read_b96 (12-byte nodes)} -> 3 compares -> select one of four.
// Random tables, random walks (the data decides nothing here: the instruction mix and the dependency do).
// Prints s_memtime ticks per PAIR of levels.  Build: hipcc --offload-arch=gfx950 -O3 -o lds_two_level lds_two_level.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef unsigned long long u64;
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef unsigned v3u __attribute__((ext_vector_type(3)));
typedef __attribute__((address_space(3))) unsigned short lds_u16;
typedef __attribute__((address_space(3))) u64 lds_u64;
typedef __attribute__((address_space(3))) v4u lds_u4;
typedef __attribute__((address_space(3))) v3u lds_u3;
typedef __attribute__((address_space(3))) unsigned lds_u32;
#define LDS_AT(type, a) (reinterpret_cast<type *>((__UINTPTR_TYPE__)(unsigned)(a)))

// LDS: [0, TILE) a rank tile [529][64] u16 (67 712 bytes), [IMG, IMG + IMGB) the tree image
constexpr int TILE = 529 * 128, IMG = 69632, IMGB = 90112;   // 88 KiB of nodes, like a group of configs[4]

// MODE 0: today's level (wide word: feature 10 | pair index 11 | rank 11; children adjacent, one b64)
// MODE 1: super-node of 16 bytes {w0 = f0|r0, w1 = fL|rL, w2 = fR|rR (10-bit feature, 11-bit rank each), w3 = quad index}
// MODE 2: super-node of 12 bytes (the quad index shares w0..w2's spare bits): ds_read_b96
template <int MODE>
__device__ __forceinline__ void pair_of_levels(v4u &cur, unsigned lk)
{
    if (MODE == 0) {
#pragma unroll
        for (int l = 0; l < 2; l++) {
            const unsigned w = cur.x;
            const unsigned xa = ((w & 0x3FFu) << 7) + lk;
            unsigned t;
            asm("v_bfe_u32 %0, %1, 10, 11" : "=v"(t) : "v"(w));
            const unsigned xv = *LDS_AT(const lds_u16, xa);
            const u64 pr = *LDS_AT(const lds_u64, IMG + (t << 3));
            cur.x = xv <= (w >> 21) ? (unsigned)pr : (unsigned)(pr >> 32);
        }
    } else {
        const unsigned x0 = *LDS_AT(const lds_u16, ((cur.x & 0x3FFu) << 7) + lk);
        const unsigned xl = *LDS_AT(const lds_u16, ((cur.y & 0x3FFu) << 7) + lk);
        const unsigned xr = *LDS_AT(const lds_u16, ((cur.z & 0x3FFu) << 7) + lk);
        v4u q[4];
        if (MODE == 1) {
            const unsigned qa = IMG + ((cur.w & 0x3FFu) << 6);   // 1 024 quads of 64 bytes
#pragma unroll
            for (int k = 0; k < 4; k++) q[k] = *LDS_AT(const lds_u4, qa + 16 * k);
        } else {
            unsigned t;  // quad index from the spare top bits of w0 (10 + 11 used: 11 spare)
            asm("v_bfe_u32 %0, %1, 21, 11" : "=v"(t) : "v"(cur.x));
            const unsigned qa = IMG + t * 48;                      // quads of 4 x 12 bytes
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const v3u p = *LDS_AT(const lds_u3, qa + 12 * k);
                q[k] = v4u{p.x, p.y, p.z, 0u};
            }
        }
        const bool g0 = x0 <= ((cur.x >> 10) & 0x7FFu), gl = xl <= ((cur.y >> 10) & 0x7FFu), gr = xr <= ((cur.z >> 10) & 0x7FFu);
        const v4u a = gl ? q[0] : q[1], b = gr ? q[2] : q[3];
        cur = g0 ? a : b;
    }
}

template <int MODE>
__global__ void chase(const unsigned *init, int pairs, int reps, long long *out, unsigned *sink)
{
    extern __shared__ char lds[];
    const int tid = threadIdx.x, nthr = blockDim.x;
    for (int i = tid; i < TILE / 4; i += nthr) *LDS_AT(lds_u32, 4 * i) = (i * 2654435761u) & 0x07FF07FFu;   // 11-bit codes
    for (int i = tid; i < IMGB / 4; i += nthr) *LDS_AT(lds_u32, IMG + 4 * i) = init[i];
    __syncthreads();
    v4u cur;
    cur.x = init[(tid * 7) & 4095];
    cur.y = init[(tid * 13 + 1) & 4095];
    cur.z = init[(tid * 29 + 2) & 4095];
    cur.w = init[(tid * 31 + 3) & 4095];
    const unsigned lk = (tid & 63) << 1;
    long long t0 = 0;
    for (int r = 0; r < reps; r++) {
        if (r == 1) t0 = __builtin_amdgcn_s_memtime();
        for (int d = pairs; d >= 2; d -= 2) {
            pair_of_levels<MODE>(cur, lk);
            pair_of_levels<MODE>(cur, lk);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (cur.x + cur.y + cur.z + cur.w == 0x12345678u) sink[0] = cur.x;
    if ((tid & 63) == 0) out[blockIdx.x * (nthr / 64) + (tid >> 6)] = t1 - t0;
}

template <int MODE>
double run(const unsigned *d_init, int waves, long long *d_out, unsigned *d_sink)
{
    const int reps = 41, grid = 256, pairs = 10;   // 20 levels
    hipFuncSetAttribute(reinterpret_cast<const void *>(chase<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, IMG + IMGB);
    hipLaunchKernelGGL((chase<MODE>), dim3(grid), dim3(64 * waves), IMG + IMGB, 0, d_init, pairs, reps, d_out, d_sink);
    hipDeviceSynchronize();
    std::vector<long long> h((size_t)grid * waves);
    hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : h) s += (double)v;
    return s / (double)h.size() / (double)((reps - 1) * pairs);
}

int main()
{
    std::vector<unsigned> init(IMGB / 4);
    srand(3);
    // every word: feature in [0, 529), 11 bits of rank / pair index, 11 more random bits (pair / quad index: all in range)
    for (auto &w : init) w = (unsigned)(rand() % 529) | ((unsigned)(rand() & 0x7FF) << 10) | ((unsigned)(rand() % 1800) << 21);
    unsigned *d_init, *d_sink;
    long long *d_out;
    hipMalloc(&d_init, init.size() * 4);
    hipMalloc(&d_out, 256 * 16 * 8);
    hipMalloc(&d_sink, 4);
    hipMemcpy(d_init, init.data(), init.size() * 4, hipMemcpyHostToDevice);
    printf("s_memtime ticks per PAIR of levels, one dependent chain per lane, random walks; rows: walking waves per CU\n");
    printf("waves   today (2 x {u16, b64})   super-node 16 B (3 x u16 + 4 x b128)   super-node 12 B (3 x u16 + 4 x b96)\n");
    const int wl[] = {1, 4, 8, 12, 14, 16};
    for (int w : wl) {
        printf("%5d   %12.1f   %22.1f   %26.1f\n", w, run<0>(d_init, w, d_out, d_sink), run<1>(d_init, w, d_out, d_sink),
               run<2>(d_init, w, d_out, d_sink));
        fflush(stdout);
    }
    return 0;
}
