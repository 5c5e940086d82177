// fcm_step_variant.hip — one instantiation of the step kernel per object file
// (-DFCM_MAXT=6|14 -DFCM_MAXNW=1|2|4), so the variants compile in parallel.
#include "fcm_kernels_common.hpp"

#ifndef FCM_MAXT
#error "compile with -DFCM_MAXT=.. -DFCM_MAXNW=.."
#endif
#define FCM_CAT2(a, b, c) a##b##_##c
#define FCM_CAT(a, b, c) FCM_CAT2(a, b, c)

// MAXT=6 variants serve the BASELINE configs: keep >= 4 waves/SIMD so that
// 4096 chains (16 waves per CU) are resident at once.
#if FCM_MAXT <= 6
#define FCM_MINW 4
#else
#define FCM_MINW 1
#endif

extern "C" int FCM_CAT(fcm_launch_step_, FCM_MAXT, FCM_MAXNW)(const FcmStepParams *p, void *stream)
{
    fcm_step_kernel<FCM_MAXT, FCM_MAXNW, FCM_MINW><<<dim3(p->nchains), dim3(WAVE), 0, (hipStream_t)stream>>>(*p);
    return (int)hipGetLastError();
}
