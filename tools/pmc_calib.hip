// pmc_calib.hip — calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE for the step kernel's access patterns
// (MI355X_MICROARCH.md "HBM": widths other than 16 B per lane are uncalibrated).  Each wave reads R random rows of a
// 2 GiB buffer, one dword per lane, and (write mode) read-modify-writes one dword in every 16th row, like the commit.
//   row_bytes = 128   lane l reads dword l & 31 of the row: the whole 128-B line (graphs of <= 1024 vertices)
//   row_bytes > 128   lane l reads the dword that holds one of 40 positions spread over the row (lane l: position
//                     (l mod 40) * row_dwords / 40, lanes >= 40 repeat): the lines a local build of 40 vertices touches in
//                     a row of 512 B (n = 4000: 4 lines) or 3840 B (n = 30000: 30 lines; with --few a build of 3
//                     vertices: 3 lines), scattered over the row like the kernel's
// Known bytes: reads = waves * R * (distinct 128-B lines touched per row) * 128, printed; writes = waves * R / 16 dwords.
// usage: pmc_calib <waves> <R> [row_bytes=128] [positions=40]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <set>

__global__ __launch_bounds__(64) void calib_kernel(uint32_t *buf, uint64_t nrows, uint32_t row_dwords, uint32_t npos, int R, uint32_t *sink, int do_write)
{
    const int lane = threadIdx.x;
    uint64_t x = (uint64_t)blockIdx.x * 0x9E3779B97F4A7C15ull + 12345;
    const uint32_t dw = row_dwords == 32u ? (uint32_t)(lane & 31) : (uint32_t)(((uint64_t)(lane % npos) * row_dwords) / npos);
    uint32_t acc = 0;
    for (int i = 0; i < R; i += 16) {
        uint32_t w[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            x = x * 6364136223846793005ull + 1442695040888963407ull;
            const uint64_t row = (x >> 20) % nrows;
            w[q] = buf[row * row_dwords + dw];
            if (do_write && q == 0 && lane == 0) buf[row * row_dwords + 5] = w[q] + 1;
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) acc ^= w[q];
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main(int argc, char **argv)
{
    const uint64_t bytes = 2ull << 30;
    const int waves = argc > 1 ? atoi(argv[1]) : 4096, R = argc > 2 ? atoi(argv[2]) : 4096;
    const uint32_t row_bytes = argc > 3 ? (uint32_t)atoi(argv[3]) : 128u, npos = argc > 4 ? (uint32_t)atoi(argv[4]) : 40u;
    const uint32_t row_dwords = row_bytes / 4;
    const uint64_t nrows = bytes / row_bytes;
    std::set<uint32_t> lines;
    for (int lane = 0; lane < 64; ++lane)
        lines.insert((row_dwords == 32u ? (uint32_t)(lane & 31) : (uint32_t)(((uint64_t)(lane % npos) * row_dwords) / npos)) / 32u);
    uint32_t *buf, *sink;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
    hipMemset(buf, 1, bytes);
    hipDeviceSynchronize();
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(calib_kernel, dim3(waves), dim3(64), 0, 0, buf, nrows, row_dwords, npos, R, sink, mode);
            hipDeviceSynchronize();
        }
    }
    printf("{\"waves\": %d, \"R\": %d, \"row_bytes\": %u, \"positions\": %u, \"lines_per_row\": %zu, \"read_bytes_per_launch\": %llu, \"write_dwords_per_launch_mode1\": %llu}\n",
           waves, R, row_bytes, npos, lines.size(), (unsigned long long)waves * R * lines.size() * 128ull, (unsigned long long)waves * (R / 16));
    return 0;
}
