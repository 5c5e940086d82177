// fcm_step_variant.hip — one instantiation of the step kernel per object file
// (-DFCM_MAXT=6|14), so the variants compile in parallel.
#include "fcm_kernels_common.hpp"

#ifndef FCM_MAXT
#error "compile with -DFCM_MAXT=6 or 14"
#endif
#define FCM_CAT2(a, b) a##b
#define FCM_CAT(a, b) FCM_CAT2(a, b)

// MAXT=6 serves the BASELINE configs (<= 8 count entries): keep >= 4 waves/SIMD
// so that 4096 chains (16 waves per CU) are resident at once.
#if FCM_MAXT <= 6
#define FCM_MINW 4
#else
#define FCM_MINW 1
#endif

extern "C" int FCM_CAT(fcm_launch_step_, FCM_MAXT)(const FcmStepParams *p, void *stream)
{
    const size_t lds = sizeof(u64) * fcm_lds_words(p->maxnw);
    fcm_step_kernel<FCM_MAXT, FCM_MINW><<<dim3(p->nchains), dim3(WAVE), lds, (hipStream_t)stream>>>(*p);
    return (int)hipGetLastError();
}
