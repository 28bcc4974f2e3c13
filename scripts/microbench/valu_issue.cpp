// valu_issue.cpp -- what one SIMD of gfx950 issues per cycle, by waves per SIMD and instruction kind.
// One workgroup per CU of 64 * 4 * W threads (W waves per SIMD), every wave runs N iterations of a block of 32
// INDEPENDENT (8 accumulators) or DEPENDENT (1 accumulator) instructions of one kind; cycles by s_memtime around the
// loop (100 MHz constant clock is s_memrealtime; s_memtime counts shader clocks), wall time by HIP events.
//   hipcc --offload-arch=gfx950 -O3 -o valu_issue valu_issue.cpp && ./valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int KIND, bool DEP>
__global__ void k(float *out, int iters, unsigned long long *cyc)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float c = 1.0f + out[0];
    float tmp = 0.0f;
    unsigned long long msk = __builtin_amdgcn_ballot_w64(a0 < 17.0f + c), msk2 = 0;
    const float sc = __builtin_amdgcn_readfirstlane(c);
    if constexpr (KIND == 16 || KIND == 24 || KIND == 26 || KIND == 28) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a0), "v"(17.0f + c) : "vcc");
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
#define OP(acc)                                                                                                  \
    if constexpr (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc) : "v"(c));                           \
    else if constexpr (KIND == 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(acc) : "v"(c));                      \
    else if constexpr (KIND == 2) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(acc) : "v"(c));             \
    else if constexpr (KIND == 3) asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(acc)); \
    else if constexpr (KIND == 4) asm volatile("v_min3_f32 %0, %0, %1, %1" : "+v"(acc) : "v"(c));                 \
    else if constexpr (KIND == 5) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(acc), "v"(c) : "vcc");          \
    else if constexpr (KIND == 6) asm volatile("s_nop 0");                                                         \
    else if constexpr (KIND == 7) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(acc) : "v"(c));                  \
    else if constexpr (KIND == 8) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(acc) : "v"(c), "s"(msk));    \
    else if constexpr (KIND == 9) asm volatile("v_cndmask_b32 %0, %1, %2, %3" : "=v"(tmp) : "v"(acc), "v"(c), "s"(msk)); \
    else if constexpr (KIND == 10) asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc) : "s"(sc));                      \
    else if constexpr (KIND == 11) asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "+v"(acc) : "v"(c)); \
    else if constexpr (KIND == 12) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(acc) : "v"(c));                  \
    else if constexpr (KIND == 13) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(acc) : "v"(c));                \
    else if constexpr (KIND == 14) asm volatile("v_cmp_lt_f32 %0, %1, %2" : "=s"(msk2) : "v"(acc), "v"(c));        \
    else if constexpr (KIND == 15) asm volatile("v_mov_b32 %0, %1" : "=v"(tmp) : "v"(acc));                          \
    else if constexpr (KIND == 16) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(acc) : "v"(c));             \
    else if constexpr (KIND == 17) asm volatile("v_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(acc) : "v"(c)); \
    else if constexpr (KIND == 18) asm volatile("v_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_cndmask_b32 %0, %0, %1, %2" : "+v"(acc) : "v"(c), "s"(msk)); \
    else if constexpr (KIND == 19) asm volatile("v_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(acc) : "v"(c) : "vcc"); \
    else if constexpr (KIND == 20) asm volatile("v_add_f32 %0, %0, %2\n\tv_add_f32 %0, %0, %2\n\tv_cmp_lt_f32 %1, %0, %2\n\ts_nop 1\n\tv_cndmask_b32 %0, %0, %2, %1" : "+v"(acc), "=s"(msk2) : "v"(c)); \
    else if constexpr (KIND == 21) asm volatile("v_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1" : "+v"(acc) : "v"(c)); \
    else if constexpr (KIND == 22) asm volatile("v_add_f32 %0, %0, %1\n\tv_min3_f32 %0, %0, %1, %1\n\tv_add_f32 %0, %0, %1\n\tv_min3_f32 %0, %0, %1, %1" : "+v"(acc) : "v"(c)); \
    else if constexpr (KIND == 23) asm volatile("v_add_f32 %0, %0, %1\n\tv_cndmask_b32 %0, %0, %1, %2\n\tv_add_f32 %0, %0, %1\n\tv_cndmask_b32 %0, %0, %1, %2" : "+v"(acc) : "v"(c), "s"(msk)); \
    else if constexpr (KIND == 24) asm volatile("v_add_f32 %0, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc\n\tv_add_f32 %0, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(acc) : "v"(c)); \
    else if constexpr (KIND == 25) asm volatile("v_min3_f32 %0, %0, %1, %1\n\tv_cndmask_b32 %0, %0, %1, %2\n\tv_min3_f32 %0, %0, %1, %1\n\tv_cndmask_b32 %0, %0, %1, %2" : "+v"(acc) : "v"(c), "s"(msk)); \
    else if constexpr (KIND == 26) asm volatile("v_min3_f32 %0, %0, %1, %1\n\tv_cndmask_b32 %0, %0, %1, vcc\n\tv_min3_f32 %0, %0, %1, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(acc) : "v"(c)); \
    else if constexpr (KIND == 27) asm volatile("v_add_f32 %0, %0, %1\n\tv_min3_f32 %0, %0, %1, %1\n\tv_cndmask_b32 %0, %0, %1, %2\n\tv_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(c), "s"(msk)); \
    else if constexpr (KIND == 28) asm volatile("v_add_f32 %0, %0, %1\n\tv_min3_f32 %0, %0, %1, %1\n\tv_cndmask_b32 %0, %0, %1, vcc\n\tv_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(c));
        if constexpr (DEP) {
#pragma unroll
            for (int u = 0; u < 32; u++) { OP(a0) }
        } else {
#pragma unroll
            for (int u = 0; u < 4; u++) { OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7) }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x + 1] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + tmp + (float)msk2;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND, bool DEP>
int run(const char *name, float *out, unsigned long long *cyc, int n_cu)
{
    const int iters = 20000;
    for (int w : {1, 2, 4, 8}) {
        const int threads = 64 * 4 * w;
        if (threads > 1024) {   // 8 waves per SIMD = two workgroups of 1 024 per CU
        }
        const int wg_threads = threads > 1024 ? 1024 : threads, grid = n_cu * (threads / wg_threads);
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        k<KIND, DEP><<<grid, wg_threads>>>(out, 10, cyc);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        k<KIND, DEP><<<grid, wg_threads>>>(out, iters, cyc);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> h(grid * (wg_threads / 64));
        CK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
        double s = 0;
        for (auto v : h) s += (double)v;
        s /= h.size();
        const double insts_per_wave = 32.0 * iters * (KIND >= 17 ? 4 : 1);
        // cycles per instruction seen by ONE wave, and per instruction of the SIMD (w waves share it)
        printf("%-14s %s waves/SIMD %d: %.2f shader cycles per instruction per wave, %.2f per SIMD instruction; wall %.3f ms -> %.2f GHz-cycles per SIMD instruction at 2.4 GHz\n",
               name, DEP ? "dependent  " : "independent", w, s / insts_per_wave, s / insts_per_wave / w, ms,
               ms * 1e-3 * 2.4e9 / (insts_per_wave * w));
    }
    return 0;
}

int main(int argc, char **)
{
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int n_cu = p.multiProcessorCount;
    printf("%s, %d CUs, clock %d kHz\n", p.name, n_cu, p.clockRate);
    float *out; unsigned long long *cyc;
    CK(hipMalloc(&out, (size_t)(n_cu * 2 * 1024 + 8) * 4));
    CK(hipMemset(out, 0, (size_t)(n_cu * 2 * 1024 + 8) * 4));
    CK(hipMalloc(&cyc, (size_t)n_cu * 2 * 16 * 8));
    run<0, false>("v_add_f32", out, cyc, n_cu);
    run<0, true>("v_add_f32", out, cyc, n_cu);
    run<1, false>("v_add_u32", out, cyc, n_cu);
    run<2, false>("v_cndmask", out, cyc, n_cu);
    run<3, false>("v_mov_dpp", out, cyc, n_cu);
    run<3, true>("v_mov_dpp", out, cyc, n_cu);
    run<4, false>("v_min3_f32", out, cyc, n_cu);
    run<5, false>("v_cmp->vcc", out, cyc, n_cu);
    run<6, false>("s_nop 0", out, cyc, n_cu);
    run<7, false>("v_fma_f32", out, cyc, n_cu);
    run<8, false>("cndmask sgpr", out, cyc, n_cu);
    run<9, false>("cndmask sgpr!", out, cyc, n_cu);
    run<16, false>("cndmask vcc=", out, cyc, n_cu);
    run<10, false>("v_add_f32 s", out, cyc, n_cu);
    run<11, false>("v_add sdwa", out, cyc, n_cu);
    run<12, false>("v_add3_u32", out, cyc, n_cu);
    run<13, false>("v_lshl_or", out, cyc, n_cu);
    run<14, false>("v_cmp->sgpr", out, cyc, n_cu);
    run<15, false>("v_mov_b32", out, cyc, n_cu);
    printf("blocks of FOUR instructions (figures are per block / 4):\n");
    run<21, false>("4 add", out, cyc, n_cu);
    run<17, false>("3add+cnd vcc", out, cyc, n_cu);
    run<18, false>("3add+cnd sgpr", out, cyc, n_cu);
    run<19, false>("2add+cmp+cnd vcc", out, cyc, n_cu);
    printf("half of the instructions of the 8-byte kinds (what a select with its mask in vcc instead of an SGPR pair is worth there):\n");
    run<22, false>("add min3 x2", out, cyc, n_cu);
    run<23, false>("add cnd-sgpr x2", out, cyc, n_cu);
    run<24, false>("add cnd-vcc x2", out, cyc, n_cu);
    run<25, false>("min3 cnd-sgpr x2", out, cyc, n_cu);
    run<26, false>("min3 cnd-vcc x2", out, cyc, n_cu);
    run<27, false>("add min3 cnd-sgpr dpp", out, cyc, n_cu);
    run<28, false>("add min3 cnd-vcc dpp", out, cyc, n_cu);
    if (argc > 1) return 0;
    run<20, false>("2add+cmp+nop+cnd s", out, cyc, n_cu);
    return 0;
}
