// Issue-rate census for gfx950: cycles per wave-instruction of the instruction kinds the step kernels are made of, by waves
// per SIMD (1, 2, 4, 8).  One workgroup of 256 * w threads per CU (256 workgroups), every wave runs REPS x 64 independent
// instructions of one kind and stamps s_memtime around the loop; printed: the median over waves of cycles / instruction,
// the slowest wave's time / (instructions of the SIMD's w waves) = what the SIMD sustains, and what the first wave to finish saw (the
// arbiter serves the oldest wave first).  EXEC variants: the same v_add with 48, 32, 16
// lanes enabled -- does the hardware skip a quarter of the wave that is switched off?
// build: hipcc --offload-arch=gfx950 -O3 -o tools/_stamp/valu_rate tools/valu_rate.hip     run: tools/_stamp/valu_rate
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#include <cstdint>

#define REPS 2000

#define R8(x) x x x x x x x x
#define R64(x) R8(R8(x))

template <int KIND>
__global__ void __launch_bounds__(1024) rate_kernel(unsigned long long *out, unsigned *sink, unsigned *gate, unsigned nblocks)
{
    unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    unsigned long long b0 = a0, b1 = a1, b2 = a2, b3 = a3;
    unsigned s0 = 1, s1 = 2, s2 = 3, s3 = 4, c1 = 0, c2 = 0;
    unsigned long long em = ~0ull;
    if (KIND == 20) em = 0x0000FFFFFFFFFFFFull;
    if (KIND == 21) em = 0x00000000FFFFFFFFull;
    if (KIND == 22) em = 0x000000000000FFFFull;
    if (KIND == 23) em = 0x0000FFFF0000FFFFull;
    __shared__ unsigned long long lds[2048];
    lds[threadIdx.x] = threadIdx.x; lds[threadIdx.x + 1024] = 1;
    const unsigned ldsa = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned long long *)&lds[threadIdx.x & 1023];
    // every workgroup of the launch resident before anybody starts (two workgroups of 1024 per CU for w = 8); gives up after a while
    if (threadIdx.x == 0) {
        atomicAdd(gate, 1u);
        for (int spin = 0; spin < 2000000 && __hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nblocks; ++spin) __builtin_amdgcn_s_sleep(8);
    }
    __syncthreads();
    unsigned long long saved;
    asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %1" : "=&s"(saved) : "s"(em));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < REPS; ++r) {
        if constexpr (KIND == 0 || (KIND >= 20 && KIND < 30)) {
            asm volatile(R8("v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %2, %2, %4\n\tv_add_u32 %3, %3, %4\n\t"
                            "v_add_u32 %4, %4, %0\n\tv_add_u32 %5, %5, %0\n\tv_add_u32 %6, %6, %0\n\tv_add_u32 %7, %7, %0\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if constexpr (KIND == 1) {   // 64-bit shift
            asm volatile(R8("v_lshlrev_b64 %0, 1, %0\n\tv_lshlrev_b64 %1, 1, %1\n\tv_lshlrev_b64 %2, 1, %2\n\tv_lshlrev_b64 %3, 1, %3\n\t"
                            "v_lshrrev_b64 %0, 1, %0\n\tv_lshrrev_b64 %1, 1, %1\n\tv_lshrrev_b64 %2, 1, %2\n\tv_lshrrev_b64 %3, 1, %3\n\t")
                         : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3));
        } else if constexpr (KIND == 2) {   // 32-bit multiply
            asm volatile(R8("v_mul_lo_u32 %0, %0, %4\n\tv_mul_lo_u32 %1, %1, %4\n\tv_mul_lo_u32 %2, %2, %4\n\tv_mul_lo_u32 %3, %3, %4\n\t"
                            "v_mul_hi_u32 %4, %4, %0\n\tv_mul_hi_u32 %5, %5, %0\n\tv_mul_hi_u32 %6, %6, %0\n\tv_mul_hi_u32 %7, %7, %0\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if constexpr (KIND == 3) {   // v_readlane into SGPRs
            asm volatile(R8("v_readlane_b32 %0, %4, 3\n\tv_readlane_b32 %1, %5, 7\n\tv_readlane_b32 %2, %6, 11\n\tv_readlane_b32 %3, %7, 13\n\t"
                            "v_readlane_b32 %0, %5, 3\n\tv_readlane_b32 %1, %6, 7\n\tv_readlane_b32 %2, %7, 11\n\tv_readlane_b32 %3, %4, 13\n\t")
                         : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
        } else if constexpr (KIND == 4) {   // bit-field extract + shift-or (what a build does per row)
            asm volatile(R8("v_bfe_u32 %0, %4, 3, 1\n\tv_lshl_or_b32 %1, %0, 5, %1\n\tv_bfe_u32 %2, %5, 3, 1\n\tv_lshl_or_b32 %3, %2, 5, %3\n\t"
                            "v_bfe_u32 %0, %6, 3, 1\n\tv_lshl_or_b32 %1, %0, 6, %1\n\tv_bfe_u32 %2, %7, 3, 1\n\tv_lshl_or_b32 %3, %2, 6, %3\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5), "v"(a6), "v"(a7));
        } else if constexpr (KIND == 5) {   // compare into an SGPR pair
            asm volatile(R8("v_cmp_ne_u32 s[20:21], %0, %1\n\tv_cmp_ne_u32 s[22:23], %1, %2\n\tv_cmp_ne_u32 s[24:25], %2, %3\n\tv_cmp_ne_u32 s[26:27], %3, %0\n\t"
                            "v_cmp_lt_u32 s[20:21], %0, %1\n\tv_cmp_lt_u32 s[22:23], %1, %2\n\tv_cmp_lt_u32 s[24:25], %2, %3\n\tv_cmp_lt_u32 s[26:27], %3, %0\n\t")
                         : : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
        } else if constexpr (KIND == 6) {   // DPP move-add (wave sums)
            asm volatile(R8("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_u32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                            "v_add_u32_dpp %2, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xf\n\tv_add_u32_dpp %3, %3, %3 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
                            "v_add_u32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_u32_dpp %5, %5, %5 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                            "v_add_u32_dpp %6, %6, %6 row_shr:4 row_mask:0xf bank_mask:0xf\n\tv_add_u32_dpp %7, %7, %7 row_shr:8 row_mask:0xf bank_mask:0xf\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if constexpr (KIND == 7) {   // popcount / find-first
            asm volatile(R8("v_bcnt_u32_b32 %0, %4, %0\n\tv_ffbl_b32 %1, %5\n\tv_bcnt_u32_b32 %2, %6, %2\n\tv_ffbl_b32 %3, %7\n\t"
                            "v_bcnt_u32_b32 %0, %5, %0\n\tv_ffbl_b32 %1, %6\n\tv_bcnt_u32_b32 %2, %7, %2\n\tv_ffbl_b32 %3, %4\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5), "v"(a6), "v"(a7));
        } else if constexpr (KIND == 8) {   // scalar ALU only
            asm volatile(R8("s_add_u32 %0, %0, %1\n\ts_and_b32 %1, %1, %2\n\ts_lshl_b32 %2, %2, 1\n\ts_xor_b32 %3, %3, %0\n\t"
                            "s_add_u32 %0, %0, %3\n\ts_or_b32 %1, %1, %2\n\ts_lshr_b32 %2, %2, 1\n\ts_xor_b32 %3, %3, %1\n\t")
                         : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
        } else if constexpr (KIND == 9) {   // vector and scalar interleaved 1 : 1 (64 + 64 per pass: counted as 64)
            asm volatile(R8("v_add_u32 %0, %0, %4\n\ts_add_u32 %8, %8, %9\n\tv_add_u32 %1, %1, %4\n\ts_and_b32 %9, %9, %10\n\tv_add_u32 %2, %2, %4\n\ts_lshl_b32 %10, %10, 1\n\tv_add_u32 %3, %3, %4\n\ts_xor_b32 %11, %11, %8\n\t"
                            "v_add_u32 %4, %4, %0\n\ts_add_u32 %8, %8, %11\n\tv_add_u32 %5, %5, %0\n\ts_or_b32 %9, %9, %10\n\tv_add_u32 %6, %6, %0\n\ts_lshr_b32 %10, %10, 1\n\tv_add_u32 %7, %7, %0\n\ts_xor_b32 %11, %11, %9\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
        } else if constexpr (KIND == 10) {  // 64-bit and / or as the compiler emits them: two 32-bit instructions -- here v_and_b32 pairs
            asm volatile(R8("v_and_b32 %0, %0, %4\n\tv_or_b32 %1, %1, %4\n\tv_and_b32 %2, %2, %4\n\tv_or_b32 %3, %3, %4\n\t"
                            "v_xor_b32 %4, %4, %0\n\tv_and_or_b32 %5, %5, %0, %1\n\tv_xor_b32 %6, %6, %0\n\tv_and_or_b32 %7, %7, %0, %2\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if constexpr (KIND == 11) {  // v_writelane
            asm volatile(R8("v_writelane_b32 %0, %4, 3\n\tv_writelane_b32 %1, %5, 7\n\tv_writelane_b32 %2, %6, 11\n\tv_writelane_b32 %3, %7, 13\n\t"
                            "v_writelane_b32 %0, %5, 4\n\tv_writelane_b32 %1, %6, 8\n\tv_writelane_b32 %2, %7, 12\n\tv_writelane_b32 %3, %4, 14\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(s0), "s"(s1), "s"(s2), "s"(s3));
        } else if constexpr (KIND == 12) {  // dependent chain of v_add (latency, one wave)
            asm volatile(R64("v_add_u32 %0, %0, %1\n\t") : "+v"(a0) : "v"(a1));
        } else if constexpr (KIND == 13) {  // dependent chain of scalar adds
            asm volatile(R64("s_add_u32 %0, %0, %1\n\t") : "+s"(s0) : "s"(s1) : "scc");
        } else if constexpr (KIND == 30 || KIND == 31 || KIND == 32) {
            // the headline kernel's mix per proposal, scaled by 1/36: 20 vector, 16 scalar, 4 branches, 4 LDS reads (KIND 31: without the
            // scalar instructions; 32: vector only) -- x 8 = 160 vector instructions per pass (counted as 64 below: multiply by 2.5)
            asm volatile(R8(
                "v_add_u32 %0, %0, %4\n\ts_add_u32 %8, %8, %9\n\tv_and_b32 %1, %1, %4\n\tv_bfe_u32 %2, %5, 3, 1\n\ts_and_b32 %9, %9, %10\n\tv_lshl_or_b32 %3, %2, 5, %3\n\t"
                "v_readlane_b32 %10, %4, 3\n\ts_lshl_b32 %11, %11, 1\n\tv_cmp_ne_u32 vcc, %0, %1\n\ts_xor_b32 %8, %8, %11\n\tv_lshlrev_b64 %[b0], 1, %[b0]\n\tds_read_b32 %[b1], %[la]\n\t"
                "s_cmp_lg_u32 %9, 0\n\ts_cbranch_scc0 1f\n\tv_add_u32 %4, %4, %0\n\t1:\n\tv_or_b32 %5, %5, %0\n\ts_add_u32 %9, %9, %8\n\tv_xor_b32 %6, %6, %0\n\ts_bfe_u32 %11, %8, 0x10003\n\t"
                "v_readlane_b32 %8, %5, 7\n\ts_or_b32 %9, %9, 1\n\tv_bcnt_u32_b32 %7, %6, %7\n\ts_lshr_b32 %11, %11, 1\n\tv_and_b32 %0, %0, %7\n\ts_waitcnt lgkmcnt(0)\n\tv_add_u32 %1, %1, %[b1]\n\t"
                "s_add_u32 %8, %8, 3\n\tv_lshrrev_b64 %[b0], 1, %[b0]\n\ts_and_b32 %10, %10, %9\n\tv_ffbl_b32 %2, %6\n\ts_cmp_eq_u32 %10, 77\n\ts_cbranch_scc1 2f\n\tv_add_u32 %3, %3, %2\n\t2:\n\t"
                "v_and_b32 %4, %4, %3\n\ts_xor_b32 %10, %10, %8\n\tv_add_u32 %5, %5, %4\n\ts_add_u32 %11, %11, %10\n\tv_xor_b32 %6, %6, %5\n\ts_and_b32 %8, %8, 0xff\n\t"
                "ds_read_b32 %[b2], %[la] offset:8\n\tv_add_u32 %7, %7, %6\n\ts_lshl_b32 %9, %9, 1\n\tv_and_b32 %0, %0, %7\n\ts_waitcnt lgkmcnt(0)\n\tv_xor_b32 %1, %1, %[b2]\n\t"
                "s_cmp_lg_u32 %11, 5\n\ts_cbranch_scc0 3f\n\tv_add_u32 %2, %2, %1\n\t3:\n\ts_or_b32 %10, %10, 2\n\tv_add_u32 %3, %3, %2\n\ts_cmp_eq_u32 %8, 99\n\ts_cbranch_scc1 4f\n\tv_xor_b32 %4, %4, %3\n\t4:\n\t")
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), [b0] "+v"(b0), [b1] "+v"(c1), [b2] "+v"(c2)
                : [la] "v"(ldsa) : "scc", "vcc", "memory");
        } else if constexpr (KIND == 14) {  // v_cndmask with an SGPR-pair condition
            asm volatile("s_mov_b64 vcc, 0x5555\n\t" R8("v_cndmask_b32 %0, %0, %4, vcc\n\tv_cndmask_b32 %1, %1, %4, vcc\n\tv_cndmask_b32 %2, %2, %4, vcc\n\tv_cndmask_b32 %3, %3, %4, vcc\n\t"
                            "v_cndmask_b32 %4, %4, %0, vcc\n\tv_cndmask_b32 %5, %5, %0, vcc\n\tv_cndmask_b32 %6, %6, %0, vcc\n\tv_cndmask_b32 %7, %7, %0, vcc\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "vcc");
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_mov_b64 exec, %0" : : "s"(saved));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (unsigned)(b0 + b1 + b2 + b3) + s0 + s1 + s2 + s3 + c1 + c2 == 0x12345u) sink[0] = 1;
}

template <int KIND>
static void run(const char *name, unsigned long long *d_out, unsigned *d_sink)
{
    printf("%-34s", name);
    for (int w : {1, 2, 4, 8}) {
        hipMemset(d_sink + 1, 0, 4);
        if (w * 256 > 1024) {   // 8 waves per SIMD: two workgroups of 1024 per CU
            hipLaunchKernelGGL(rate_kernel<KIND>, dim3(512), dim3(1024), 0, 0, d_out, d_sink, d_sink + 1, 512u);
        } else {
            hipLaunchKernelGGL(rate_kernel<KIND>, dim3(256), dim3(256 * w), 0, 0, d_out, d_sink, d_sink + 1, 256u);
        }
        hipDeviceSynchronize();
        const int nw = w == 8 ? 512 * 16 : 256 * 4 * w;
        std::vector<unsigned long long> h(nw);
        hipMemcpy(h.data(), d_out, nw * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        const double per = (double)h[nw / 2] / ((double)REPS * 64.0);
        printf("  w=%d: %6.2f (SIMD %5.2f; first %5.2f)", w, per, (double)h[nw - 1] / ((double)REPS * 64.0 * w), (double)h[0] / ((double)REPS * 64.0));
    }
    printf("\n");
}

int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    unsigned long long *d_out; unsigned *d_sink;
    hipMalloc(&d_out, 8192 * sizeof(unsigned long long));
    hipMalloc(&d_sink, 8);
    printf("cycles (s_memtime) per wave-instruction as one wave sees it, and per SIMD; w = waves per SIMD\n");
    run<0>("v_add_u32", d_out, d_sink);
    run<10>("v_and/or/xor/and_or_b32", d_out, d_sink);
    run<1>("v_lshl/lshrrev_b64", d_out, d_sink);
    run<2>("v_mul_lo/hi_u32", d_out, d_sink);
    run<3>("v_readlane_b32", d_out, d_sink);
    run<11>("v_writelane_b32", d_out, d_sink);
    run<4>("v_bfe_u32 + v_lshl_or_b32", d_out, d_sink);
    run<5>("v_cmp -> SGPR pair", d_out, d_sink);
    run<14>("v_cndmask_b32 vcc", d_out, d_sink);
    run<6>("v_add_u32 dpp row_shr", d_out, d_sink);
    run<7>("v_bcnt / v_ffbl", d_out, d_sink);
    run<8>("scalar ALU", d_out, d_sink);
    run<9>("v_add + scalar 1:1 (per pair)", d_out, d_sink);
    run<12>("v_add_u32 dependent chain", d_out, d_sink);
    run<13>("s_add_u32 dependent chain", d_out, d_sink);
    run<30>("mix 26v+19s+4br+2lds per 8 (x 1/6.4)", d_out, d_sink);
    run<20>("v_add_u32, 48 lanes enabled", d_out, d_sink);
    run<21>("v_add_u32, 32 lanes enabled", d_out, d_sink);
    run<22>("v_add_u32, 16 lanes enabled", d_out, d_sink);
    run<23>("v_add_u32, lanes 0-15 + 32-47", d_out, d_sink);
    return 0;
}
