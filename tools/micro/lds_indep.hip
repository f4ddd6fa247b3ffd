// GPU-box microbenchmark (diagnostic, not part of the product): what ONE LDS wave-instruction
// costs a CU when the reads are INDEPENDENT (nothing chained), by instruction kind, waves per
// CU, reads in flight per wave and address pattern -- and the same for the forest walk's own
// level (ds_read_u16 + ds_read_b64 + 5 VALU, dependent) by waves x chains per lane.
// Settles EXPERIMENTS.md 4.2's "4 cycles per LDS instruction" against MI355X_MICROARCH.md's LDS
// table (ds_read_b32 / b64: 2 LDS-array cycles).
// Build: hipcc --offload-arch=gfx950 -O3 -o lds_indep lds_indep.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef unsigned long long u64;
typedef __attribute__((address_space(3))) unsigned lds_u32;
typedef __attribute__((address_space(3))) u64 lds_u64;
#define LDS_AT(type, a) (reinterpret_cast<type *>((__UINTPTR_TYPE__)(unsigned)(a)))

constexpr int LDS_BYTES = 131072;

// KIND 0: ds_read_u16   1: ds_read_b32   2: ds_read_b64   3: u16 + b64 alternating (the walk's mix)
//      4: ds_read_u16_d16 + ds_read_u16_d16_hi pairs (two codes into one register)
// PAT  0: every lane its own bank (conflict-free)   1: all lanes one address (broadcast)
//      2: pseudo-random addresses (bank conflicts as they fall)
// NV: independent VALU instructions issued per read (issue-port competition)
template <int KIND, int K, int NV>
__global__ void indep(int reps, int pat, long long *out, unsigned *sink)
{
    extern __shared__ char lds[];
    const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63;
    for (int i = tid; i < LDS_BYTES / 4; i += nthr) *LDS_AT(lds_u32, 4 * i) = i * 2654435761u;
    __syncthreads();
    unsigned a16[K], a64[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
        unsigned r = (tid * 2654435761u + k * 40503u) >> 7;
        a16[k] = pat == 0 ? lane * 4 + k * 256 : pat == 1 ? k * 256 : (r & 0x7ffe);
        a64[k] = 32768 + (pat == 0 ? lane * 8 + k * 512 : pat == 1 ? k * 512 : (r & 0xfff8));
    }
    unsigned acc = 0, vv[4] = {1u, 2u, 3u, 4u};
    long long t0 = 0;
    for (int r = 0; r < reps; r++) {
        if (r == 1) t0 = __builtin_amdgcn_s_memtime();
        unsigned x[K];
        u64 p[K];
#pragma unroll
        for (int k = 0; k < K; k++) {
            if (KIND == 0) asm volatile("ds_read_u16 %0, %1" : "=v"(x[k]) : "v"(a16[k]));
            if (KIND == 1) asm volatile("ds_read_b32 %0, %1" : "=v"(x[k]) : "v"(a16[k]));
            if (KIND == 2) asm volatile("ds_read_b64 %0, %1" : "=v"(p[k]) : "v"(a64[k]));
            if (KIND == 3) {
                asm volatile("ds_read_u16 %0, %1" : "=v"(x[k]) : "v"(a16[k]));
                asm volatile("ds_read_b64 %0, %1" : "=v"(p[k]) : "v"(a64[k]));
            }
            if (KIND == 4) {
                x[k] = 0;
                asm volatile("ds_read_u16_d16 %0, %1" : "+v"(x[k]) : "v"(a16[k]));
                asm volatile("ds_read_u16_d16_hi %0, %1 offset:2" : "+v"(x[k]) : "v"(a16[k]));
            }
#pragma unroll
            for (int j = 0; j < NV; j++) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(vv[j & 3]) : "v"(acc));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < K; k++) {
            if (KIND != 2) acc ^= x[k];
            if (KIND == 2 || KIND == 3) acc ^= (unsigned)p[k] ^ (unsigned)(p[k] >> 32);
        }
        // (addresses stay what they were: the next trip's reads do not depend on this one's)
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if ((acc ^ vv[0] ^ vv[1] ^ vv[2] ^ vv[3]) == 0x12345678u) sink[0] = acc;
    if (lane == 0) out[blockIdx.x * (nthr / 64) + (tid >> 6)] = t1 - t0;
}

// The rank walk's own level, dependent: word w = [31:21 rank | 20 | 19:8 pair | 7:0 feature];
// codes at [0, 32768): [F][64][2] u16, pairs at 32768 + 8 * pair.  CH chains per lane.
// VARIANT 0: as pk_forest_q.hip (perm, bfe, lshl_add, u16 read, b64 read, cmp, cndmask)
// VARIANT 1: the code read as ds_read_b32 of the lane's dword (both candidates' codes), the half
//            picked by a v_bfe / shift (tests whether u16 costs more than b32)
template <int CH, int VARIANT>
__global__ void walk(const unsigned *init, int levels, int reps, int coherent, long long *out, unsigned *sink)
{
    extern __shared__ char lds[];
    const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63;
    for (int i = tid; i < 32768 / 4; i += nthr)
        *LDS_AT(lds_u32, 4 * i) = coherent ? 0u : ((i * 2654435761u) & 0xffe0ffe0u);
    for (int i = tid; i < 98304 / 4; i += nthr) *LDS_AT(lds_u32, 32768 + 4 * i) = init[i];
    __syncthreads();
    unsigned w[CH];
#pragma unroll
    for (int c = 0; c < CH; c++) w[c] = init[coherent ? 2 * c : ((tid * 7 + c * 131) % 24576)];
    const unsigned lk0 = lane << 2, lk1 = lk0 + 2;
    long long t0 = 0;
    for (int r = 0; r < reps; r++) {
        if (r == 1) t0 = __builtin_amdgcn_s_memtime();
        for (int d = levels; d >= 2; d -= 2) {
#pragma unroll
            for (int rep = 0; rep < 2; rep++) {
                unsigned xv[CH];
                u64 pr[CH];
#pragma unroll
                for (int c = 0; c < CH; c++) {
                    const unsigned xa = __builtin_amdgcn_perm(w[c], (c & 1) ? lk1 : lk0, 0x0c0c0400u);
                    if (VARIANT == 0) xv[c] = *LDS_AT(const volatile __attribute__((address_space(3))) unsigned short, xa);
                    else {
                        const unsigned both = *LDS_AT(const volatile lds_u32, xa & ~3u);
                        xv[c] = (c & 1) ? both >> 16 : both & 0xffffu;
                    }
                    unsigned t;
                    asm("v_bfe_u32 %0, %1, 8, 12" : "=v"(t) : "v"(w[c]));
                    pr[c] = *LDS_AT(const volatile lds_u64, 32768u + (t << 3));
                }
#pragma unroll
                for (int c = 0; c < CH; c++) {
                    const bool gl = xv[c] <= (w[c] >> 16);
                    w[c] = gl ? (unsigned)pr[c] : (unsigned)(pr[c] >> 32);
                }
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    unsigned s = 0;
#pragma unroll
    for (int c = 0; c < CH; c++) s += w[c];
    if (s == 0x12345678u) sink[0] = s;
    if (lane == 0) out[blockIdx.x * (nthr / 64) + (tid >> 6)] = t1 - t0;
}

static long long *d_out;
static unsigned *d_sink, *d_init;

static double mean_ticks(int grid, int waves)
{
    hipDeviceSynchronize();
    std::vector<long long> h((size_t)grid * waves);
    hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : h) s += (double)v;
    return s / (double)h.size();
}

template <int KIND, int K, int NV>
double run_indep(int waves, int pat)
{
    const int reps = 201, grid = 256;
    hipFuncSetAttribute(reinterpret_cast<const void *>(indep<KIND, K, NV>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipLaunchKernelGGL((indep<KIND, K, NV>), dim3(grid), dim3(64 * waves), LDS_BYTES, 0, reps, pat, d_out, d_sink);
    const int per_k = (KIND == 3 || KIND == 4) ? 2 : 1;
    // CU cycles per LDS wave-instruction: the waves run concurrently, so the CU issued
    // waves * K * per_k instructions per trip in (ticks / trips) cycles
    return mean_ticks(grid, waves) / (double)(reps - 1) / (double)(waves * K * per_k);
}

template <int CH, int VARIANT>
double run_walk(int waves, int coherent)
{
    const int reps = 21, levels = 20, grid = 256;
    hipFuncSetAttribute(reinterpret_cast<const void *>(walk<CH, VARIANT>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipLaunchKernelGGL((walk<CH, VARIANT>), dim3(grid), dim3(64 * waves), LDS_BYTES, 0, d_init, levels, reps,
                       coherent, d_out, d_sink);
    // CU cycles per walk-level (one candidate-walk of one wave = 2 LDS instructions)
    return mean_ticks(grid, waves) / (double)((reps - 1) * levels) / (double)(waves * CH);
}

template <int KIND, int NV>
void indep_rows(const char *name)
{
    const int wl[] = {4, 8, 16};
    for (int pat = 0; pat < 3; pat++)
        for (int w : wl)
            printf("%-22s pat=%d waves=%2d  K=1 %6.2f  K=2 %6.2f  K=4 %6.2f  K=8 %6.2f  K=16 %6.2f\n", name, pat, w,
                   run_indep<KIND, 1, NV>(w, pat), run_indep<KIND, 2, NV>(w, pat), run_indep<KIND, 4, NV>(w, pat),
                   run_indep<KIND, 8, NV>(w, pat), run_indep<KIND, 16, NV>(w, pat));
    fflush(stdout);
}

int main()
{
    std::vector<unsigned> init(98304 / 4);
    srand(1);
    for (size_t i = 0; i < init.size(); i++) {
        // 12288 pairs of 8 bytes; a word: rank (11 bits) | pair (12 bits, < 4096: 32 KiB of pairs
        // in reach, like a staged group's hot part) | feature (< 121)
        const unsigned pair = (unsigned)(rand() % 4096), feat = (unsigned)(rand() % 121);
        init[i] = ((unsigned)(rand() % 2048) << 21) | (pair << 8) | feat;
    }
    hipMalloc(&d_init, init.size() * 4);
    hipMalloc(&d_out, 256 * 16 * 8);
    hipMalloc(&d_sink, 4);
    hipMemcpy(d_init, init.data(), init.size() * 4, hipMemcpyHostToDevice);
    printf("A. independent reads: CU cycles per LDS wave-instruction (s_memtime ticks); K = reads in flight per wave\n");
    printf("   pat 0 = a bank per lane, 1 = broadcast, 2 = random\n");
    indep_rows<0, 0>("ds_read_u16");
    indep_rows<1, 0>("ds_read_b32");
    indep_rows<2, 0>("ds_read_b64");
    indep_rows<3, 0>("u16+b64");
    indep_rows<4, 0>("u16_d16+u16_d16_hi");
    indep_rows<3, 2>("u16+b64, 2 VALU/read");
    indep_rows<3, 5>("u16+b64, 5 VALU/read");
    printf("B. the rank walk's level (dependent): CU cycles per candidate-walk-level (= 2 LDS instructions)\n");
    for (int coh = 0; coh < 2; coh++) {
        const int wl[] = {4, 8, 12, 16};
        for (int w : wl) {
            printf("walk %s waves=%2d  CH=1 %6.2f  CH=2 %6.2f  CH=4 %6.2f  CH=8 %6.2f | b32-code CH=2 %6.2f CH=4 %6.2f\n",
                   coh ? "coherent" : "random  ", w, run_walk<1, 0>(w, coh), run_walk<2, 0>(w, coh),
                   run_walk<4, 0>(w, coh), run_walk<8, 0>(w, coh), run_walk<2, 1>(w, coh), run_walk<4, 1>(w, coh));
            fflush(stdout);
        }
    }
    return 0;
}
