// fcm_step_variant.hip — one instantiation of the step kernel per object file, so the variants
// compile in parallel.  -DFCM_TAG=<name> -DFCM_MAXT=<t> -DFCM_EXACT=0|1 -DFCM_CLIQUE=0|1|2 (2: clique moves incl. local sets of 257..1024 vertices):
//   EXACT=1: the tracked depth tmax (= count entries - 2) IS FCM_MAXT, a compile-time constant: the
//            clique walk has exactly that many levels and no depth checks (tags x2_0 .. x6_1);
//   EXACT=0: tmax <= FCM_MAXT at run time (tags 6_0, 14_0, 6_1, 14_1).
#if defined(FCM_PC) && FCM_PC == 3
#define MW_TBL_N 20u   // (the clique-move kernel: a shorter draw table, so that its workgroup stays within 10 KiB of LDS -- 16 per CU)
#endif
#include "fcm_kernels_common.hpp"
#include "fcm_step_mw.hpp"
#include "fcm_step_cq.hpp"

#if !defined(FCM_MAXT) || !defined(FCM_CLIQUE) || !defined(FCM_EXACT) || !defined(FCM_TAG)
#error "compile with -DFCM_TAG=.. -DFCM_MAXT=.. -DFCM_EXACT=0|1 -DFCM_CLIQUE=0|1"
#endif
#define FCM_CAT2(a, b) a##b
#define FCM_CAT(a, b) FCM_CAT2(a, b)

// Up to 6 levels serves the BASELINE configs (<= 8 count entries): keep >= 4 waves/SIMD
// (at most 128 VGPRs) so that 4096 chains (16 waves per CU) are resident at once.
#if FCM_MAXT <= 6
#define FCM_MINW 4
#else
#define FCM_MINW 1
#endif

#if defined(FCM_PC) && FCM_PC == 3
// tags c2_1 .. c6_1: the cooperative step kernel for move mixes with clique moves (fcm_step_cq.hpp): W = p->mw_waves (or 1)
// waves per chain, the pairs of a move on the pre-move bitmap, one commit on accept
extern "C" int FCM_CAT(fcm_launch_step_, FCM_TAG)(const FcmStepParams *p, void *stream)
{
    const unsigned W = p->mw_waves >= 2 ? p->mw_waves : 1u;
    const size_t words = fcm_cq_lds_words(p->maxnw < 2 ? 2 : p->maxnw, p->chg_cap, W);
#ifndef CQ_WFIX
#define CQ_WFIX 1   // 1: every W (1, 2, 4, 8) runs an instantiation with the wave count folded in; 0: one kernel, W at run time
#endif
#define CQ_LAUNCH(WF) fcm_step_cq_kernel<FCM_MAXT, WF><<<dim3(p->nchains), dim3(W * WAVE), sizeof(u64) * words, (hipStream_t)stream>>>(*p)
    if (CQ_WFIX && W == 1u) CQ_LAUNCH(1);
    else if (CQ_WFIX && W == 2u) CQ_LAUNCH(2);
    else if (CQ_WFIX && W == 4u) CQ_LAUNCH(4);
    else if (CQ_WFIX && W == 8u) CQ_LAUNCH(8);
    else CQ_LAUNCH(0);
#undef CQ_LAUNCH
    return (int)hipGetLastError();
}
#elif defined(FCM_PC) && FCM_PC
// tags m2_0 .. m6_0 (rows of one cache line, FCM_PC=1), n2_0 .. n6_0 (longer rows, FCM_PC=2) and s2_0 .. s6_0 (sparse state: two
// bits per adjacent pair instead of row bitmaps, FCM_PC=5): the
// multi-wave kernel (p->mw_waves waves per chain, in-order commit), simple moves only
extern "C" int FCM_CAT(fcm_launch_step_, FCM_TAG)(const FcmStepParams *p, void *stream)
{
    const size_t words = fcm_mw_lds_words(p->maxnw, (int)p->mw_waves);
#ifndef MW_WFIX
#define MW_WFIX 16  // every W (2, 4, 8, 16) runs an instantiation with the wave count folded in (fcm_step_mw.hpp, mw_wave); 2: only W = 2 does; 0: none
#endif
#define MW_LAUNCH(WF, WAVES) fcm_step_mw_kernel<FCM_MAXT, FCM_PC == 1, FCM_PC == 5, WF><<<dim3(p->nchains), dim3((WAVES) * WAVE), sizeof(u64) * words, (hipStream_t)stream>>>(*p)
    if (MW_WFIX >= 2 && p->mw_waves == 2u) MW_LAUNCH(2, 2);
#if MW_WFIX >= 16
    else if (p->mw_waves == 4u) MW_LAUNCH(4, 4);
    else if (p->mw_waves == 8u) MW_LAUNCH(8, 8);
    else if (p->mw_waves == 16u) MW_LAUNCH(16, 16);
#endif
    else MW_LAUNCH(0, p->mw_waves);
#undef MW_LAUNCH
    return (int)hipGetLastError();
}
#else
extern "C" int FCM_CAT(fcm_launch_step_, FCM_TAG)(const FcmStepParams *p, void *stream)
{
    size_t words = fcm_lds_words(p->maxnw);
    if (FCM_CLIQUE) words += fcm_clique_lds_words(p->chg_cap);
    words += FCM_TALLY_LDS_WORDS;
    fcm_step_kernel<FCM_MAXT, FCM_MINW, FCM_CLIQUE, FCM_EXACT != 0>
        <<<dim3(p->nchains), dim3(WAVE), sizeof(u64) * words, (hipStream_t)stream>>>(*p);
    return (int)hipGetLastError();
}
#endif
