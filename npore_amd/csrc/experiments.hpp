// experiments.hpp -- measurement instruments of the fill kernel.  NOT part of the product build: kernels.hpp /
// cell.hpp include this file only under -DNPORE_EXPERIMENTS (scripts/ab_fill.py, scripts/step_stats.py,
// tests/tools/ab_sync.py build such libraries); the default library never sees it.
//   -DNPORE_PAD_VALU=16 / NPORE_PAD_SALU=16 / NPORE_PAD_NOP=16 / NPORE_PAD_OP=1...10   extra instructions of one kind per
//                      anti-diagonal: what an instruction of that kind costs the kernel (LABNOTES.md)
//   -DNPORE_X_NOLEN / NPORE_X_NOSHR / NPORE_X_NOPOLL / NPORE_X_CHUNKMAJOR / NPORE_X_POLLSLEEP=k   ablations (wrong
//                      strings, timing only) and placement / poll variants
//   -DNPORE_STATS      per-path counters of the cell update (npore_debug_stats)
//   -DNPORE_RELAXED_SYNC   compiler barriers instead of the workgroup release / acquire fences of the hand-shakes
#pragma once
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>

#if defined(NPORE_STATS) && defined(__HIP_DEVICE_COMPILE__)
#define NPORE_COUNT(k) env.count(k)
#else
#define NPORE_COUNT(k) ((void)0)
#endif

namespace npore {

namespace xp {
#if defined(NPORE_X_NOLEN)
constexpr bool NOLEN = true;
#else
constexpr bool NOLEN = false;
#endif
#if defined(NPORE_X_NOSHR)
constexpr bool NOSHR = true;
#else
constexpr bool NOSHR = false;
#endif
#if defined(NPORE_X_NOPOLL)
constexpr bool NOPOLL = true;
#else
constexpr bool NOPOLL = false;
#endif
#if defined(NPORE_X_CHUNKMAJOR)
constexpr bool CHUNKMAJOR = true;
#else
constexpr bool CHUNKMAJOR = false;
#endif
#if defined(NPORE_X_NOASM)
constexpr bool NOASM = true;        // the plain steps through the compiled C++ body instead of fill_step_asm.inc (A/B)
#else
constexpr bool NOASM = false;
#endif
#if defined(NPORE_X_ANN)
constexpr int ANN = NPORE_X_ANN;      // annotate ablations (timing only): 1 no packing, 2 no period passes, 3 neither
#else
constexpr int ANN = 0;
#endif
#if defined(NPORE_X_PRIO)
constexpr int PRIO = NPORE_X_PRIO;    // issue priorities of a chunk's waves: 1 first wave highest, 2 last wave highest, 3 none, 4 middle > first > last
#else
constexpr int PRIO = 0;
#endif
#if defined(NPORE_X_NOPRETEST)
constexpr bool NOPRETEST = true;      // annotate: no scalar pre-test of a window's masks (A/B)
#else
constexpr bool NOPRETEST = false;
#endif
#if defined(NPORE_X_NOASM_ROLES)
constexpr int NOASM_ROLES = NPORE_X_NOASM_ROLES;      // bit k: the waves of role k run the compiled plain steps (race bisection)
#else
constexpr int NOASM_ROLES = 0;
#endif
#if defined(NPORE_X_ZEROLDS)
constexpr int ZEROLDS = NPORE_X_ZEROLDS;      // 1: the chunks' LDS zeroed at kernel start, 2: filled with NaN (uninitialised-read hunt)
#else
constexpr int ZEROLDS = 0;
#endif
#if defined(NPORE_X_DBGMAT)
constexpr bool DBGMAT = true;      // every cell's MAT.VAL to a second buffer laid out like the traceback words (race hunt)
#else
constexpr bool DBGMAT = false;
#endif
#if defined(NPORE_X_ANNT)
constexpr int ANNT = NPORE_X_ANNT;    // threads of an annotate workgroup (planes in LDS)
#else
constexpr int ANNT = 1024;
#endif
#if defined(NPORE_X_POLLSLEEP)
constexpr int POLLSLEEP = NPORE_X_POLLSLEEP;
#else
constexpr int POLLSLEEP = 0;
#endif
}  // namespace xp

#if defined(__HIPCC__)

__device__ __forceinline__ void pad_hook(uint32_t tcol4, unsigned long long lanes)
{
    (void)tcol4;
    (void)lanes;
#if defined(NPORE_PAD_VALU)
    {
        uint32_t pa = tcol4, pb = tcol4;
#pragma unroll
        for (int k = 0; k < NPORE_PAD_VALU / 2; k++) {
            asm volatile("v_add_u32 %0, %0, 1" : "+v"(pa));
            asm volatile("v_add_u32 %0, %0, 1" : "+v"(pb));
        }
    }
#endif
#if defined(NPORE_PAD_OP)
    {
        // what one more instruction of a given kind costs (16 of them, two independent chains)
        uint32_t pa = tcol4, pb = tcol4 + 1u;
        float fa = __uint_as_float(tcol4), fb = fa;
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x2 qa = {fa, fb}, qb = {fb, fa};
        unsigned long long msk = lanes;
#pragma unroll
        for (int k = 0; k < 8; k++) {
#if NPORE_PAD_OP == 1
            asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(pa));
            asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(pb));
#elif NPORE_PAD_OP == 2
            asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(pa) : "v"(pb), "s"(msk));
            asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(pb) : "v"(pa), "s"(msk));
#elif NPORE_PAD_OP == 3
            asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(qa) : "v"(qb));
            asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(qb) : "v"(qa));
#elif NPORE_PAD_OP == 4
            asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(pa) : "v"(pb));
            asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(pb) : "v"(pa));
#elif NPORE_PAD_OP == 5
            asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "+v"(pa) : "v"(pb));
            asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "+v"(pb) : "v"(pa));
#elif NPORE_PAD_OP == 6
            asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(pa));
            asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(pb));
#elif NPORE_PAD_OP == 7
            asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(msk) : "v"(fa), "v"(fb));
            asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(msk) : "v"(fb), "v"(fa));
#elif NPORE_PAD_OP == 8
            asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(pa) : "v"(pb));
            asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(pb) : "v"(pa));
#elif NPORE_PAD_OP == 9
            asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(pa) : "v"(tcol4));
            asm volatile("v_add_u32 %0, %0, 1" : "+v"(pb));
#elif NPORE_PAD_OP == 10
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(fa) : "v"(fb));
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(fb) : "v"(fa));
#endif
        }
    }
#endif
#if defined(NPORE_PAD_SALU)
    {
        uint32_t pa = 1u, pb = 2u, pc = 3u, pd = 4u;
#pragma unroll
        for (int k = 0; k < NPORE_PAD_SALU / 4; k++) {
            asm volatile("s_add_u32 %0, %0, 1" : "+s"(pa) : : "scc");
            asm volatile("s_add_u32 %0, %0, 1" : "+s"(pb) : : "scc");
            asm volatile("s_add_u32 %0, %0, 1" : "+s"(pc) : : "scc");
            asm volatile("s_add_u32 %0, %0, 1" : "+s"(pd) : : "scc");
        }
    }
#endif
#if defined(NPORE_PAD_NOP)
#pragma unroll
    for (int k = 0; k < NPORE_PAD_NOP; k++) asm volatile("s_nop 0");
#endif
}

#endif  // __HIPCC__

}  // namespace npore
