// fcm_step_variant.hip — one instantiation of the step kernel per object file
// (-DFCM_MAXT=6|14 -DFCM_CLIQUE=0|1), so the variants compile in parallel.
#include "fcm_kernels_common.hpp"

#if !defined(FCM_MAXT) || !defined(FCM_CLIQUE)
#error "compile with -DFCM_MAXT=6|14 -DFCM_CLIQUE=0|1"
#endif
#define FCM_CAT3(a, b, c, d) a##b##c##d
#define FCM_CAT(a, b, c, d) FCM_CAT3(a, b, c, d)

// MAXT=6 serves the BASELINE configs (<= 8 count entries): keep >= 4 waves/SIMD
// (at most 128 VGPRs) so that 4096 chains (16 waves per CU) are resident at once.
#if FCM_MAXT <= 6
#define FCM_MINW 4
#else
#define FCM_MINW 1
#endif

extern "C" int FCM_CAT(fcm_launch_step_, FCM_MAXT, _, FCM_CLIQUE)(const FcmStepParams *p, void *stream)
{
    size_t words = fcm_lds_words(p->maxnw);
    if (FCM_CLIQUE) words += fcm_clique_lds_words(p->chg_cap);
    fcm_step_kernel<FCM_MAXT, FCM_MINW, FCM_CLIQUE != 0><<<dim3(p->nchains), dim3(WAVE), sizeof(u64) * words, (hipStream_t)stream>>>(*p);
    return (int)hipGetLastError();
}
