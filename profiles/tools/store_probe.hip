// What does the chip sustain for a pure STORE stream (and, at the end, a pure READ stream shaped like wgrad's register staging) shaped like dgrad's gradient stash (16 B per lane, 1 KiB per wave
// instruction, nontemporal), with one 256-thread workgroup per CU (dgrad's occupancy: the register file holds no second one)
// and with more?   hipcc -O3 --offload-arch=gfx950 store_probe.hip -o /tmp/store_probe && /tmp/store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u4 __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ __launch_bounds__(256) void fill(u4* dst, size_t n_vec, int per_wave_run) {
    // each wave writes runs of `per_wave_run` consecutive 1 KiB rows, the runs of the grid's waves interleaved (the stash layout:
    // a wave owns its samples' records)
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (size_t)gridDim.x * 4;
    const int lane = threadIdx.x & 63;
    const size_t rows = n_vec / 64;
    const u4 v = {1u, 2u, 3u, (unsigned)wave};
    for (size_t r0 = wave * per_wave_run; r0 < rows; r0 += n_waves * per_wave_run)
        for (int k = 0; k < per_wave_run && r0 + k < rows; ++k) {
            if (NT) __builtin_nontemporal_store(v, dst + (r0 + k) * 64 + lane);
            else dst[(r0 + k) * 64 + lane] = v;
        }
}
__global__ __launch_bounds__(256) void copy(const u4* src, u4* dst, size_t n_vec) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_vec; i += (size_t)gridDim.x * 256)
        __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}
// a pure READ stream: every wave keeps `depth` 1 KiB rows in flight in registers (wgrad's staging), xor-folds them, one store per wave
template <int DEPTH>
__global__ __launch_bounds__(256) void readall(const u4* src, size_t n_vec, u4* sink) {
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (size_t)gridDim.x * 4;
    const int lane = threadIdx.x & 63;
    const size_t rows = n_vec / 64;
    u4 acc = {0u, 0u, 0u, 0u};
    for (size_t r0 = wave * DEPTH; r0 + DEPTH <= rows; r0 += n_waves * DEPTH) {
        u4 v[DEPTH];
#pragma unroll
        for (int k = 0; k < DEPTH; ++k) v[k] = __builtin_nontemporal_load(src + (r0 + k) * 64 + lane);
#pragma unroll
        for (int k = 0; k < DEPTH; ++k) acc ^= v[k];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) sink[wave * 64 + lane] = acc;
}
// wgrad's access pattern: the buffer is a sequence of records of `rec_kib` KiB (one per 32-sample segment); a workgroup owns a
// contiguous range of records and each of its 4 waves reads PW 1-KiB pieces of every record (slots wave + 4 pp of the `n_slots`
// the job touches, spread over the record), D records in flight
template <int PW, int D>
__global__ __launch_bounds__(256) void readpat(const char* src, size_t n_rec, int rec_kib, int slot_stride, u4* sink) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t per = (n_rec + gridDim.x - 1) / gridDim.x, r0 = blockIdx.x * per, r1 = r0 + per < n_rec ? r0 + per : n_rec;
    u4 acc = {0u, 0u, 0u, 0u};
    u4 v[D][PW];
    auto issue = [&](size_t r, int d) {
        const char* rec = src + (r < r1 ? r : r1 - 1) * (size_t)rec_kib * 1024;
#pragma unroll
        for (int pp = 0; pp < PW; ++pp)
            v[d][pp] = __builtin_nontemporal_load(reinterpret_cast<const u4*>(rec + (size_t)((wave + 4 * pp) * slot_stride) * 1024) + lane);
    };
#pragma unroll
    for (int d = 0; d < D; ++d) issue(r0 + d, d);
    for (size_t r = r0; r < r1; r += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
#pragma unroll
            for (int pp = 0; pp < PW; ++pp) acc ^= v[d][pp];
            issue(r + d + D, d);
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) sink[blockIdx.x * 256 + threadIdx.x] = acc;
}
int main() {
    const size_t bytes = 2900ull << 20;      // the fine pass' gradient stash
    u4 *a, *b;
    hipMalloc(&a, bytes);
    hipMalloc(&b, bytes);
    hipMemset(a, 0, bytes);
    hipMemset(b, 0, bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto time = [&](auto launch, const char* name, double moved) {
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-64s %7.3f ms  %6.2f TB/s\n", name, ms / 10, moved / (ms / 10 * 1e-3) / 1e12);
    };
    const size_t n = bytes / 16;
    for (int grid : {256, 512, 1024, 4096})
        for (int run : {1, 16, 192}) {
            char nm[128];
            snprintf(nm, sizeof nm, "store nt, %4d workgroups, runs of %3d KiB per wave", grid, run);
            time([&] { fill<true><<<grid, 256>>>(a, n, run); }, nm, (double)bytes);
        }
    time([&] { fill<false><<<256, 256>>>(a, n, 16); }, "store plain, 256 workgroups, runs of 16 KiB per wave", (double)bytes);
    time([&] { copy<<<4096, 256>>>(a, b, n); }, "copy nt (read + write), 4096 workgroups", 2.0 * bytes);
    time([&] { hipMemsetAsync(a, 0, bytes, 0); }, "hipMemsetAsync", (double)bytes);
    time([&] { readall<8><<<256, 256>>>(a, n, b); }, "read nt, 256 workgroups, 8 KiB in flight per wave", (double)bytes);
    time([&] { readall<16><<<256, 256>>>(a, n, b); }, "read nt, 256 workgroups, 16 KiB in flight per wave", (double)bytes);
    time([&] { readall<32><<<256, 256>>>(a, n, b); }, "read nt, 256 workgroups, 32 KiB in flight per wave", (double)bytes);
    time([&] { readall<16><<<1024, 256>>>(a, n, b); }, "read nt, 1024 workgroups, 16 KiB in flight per wave", (double)bytes);
    time([&] { readall<8><<<4096, 256>>>(a, n, b); }, "read nt, 4096 workgroups, 8 KiB in flight per wave", (double)bytes);
    {   // 2.9 GB as records of 104 KiB; a "job" reads 32 of their 104 slots (every 3rd KiB) or 32 adjacent ones
        const int rec_kib = 104;
        const size_t n_rec = bytes / ((size_t)rec_kib * 1024);
        const double moved = (double)n_rec * 32 * 1024;
        time([&] { readpat<8, 2><<<256, 256>>>((const char*)a, n_rec, rec_kib, 3, b); }, "pattern: 32 KiB of every 104 KiB record, pieces 3 KiB apart, D 2", moved);
        time([&] { readpat<8, 3><<<256, 256>>>((const char*)a, n_rec, rec_kib, 3, b); }, "pattern: the same, D 3", moved);
        time([&] { readpat<8, 4><<<256, 256>>>((const char*)a, n_rec, rec_kib, 3, b); }, "pattern: the same, D 4", moved);
        time([&] { readpat<8, 3><<<256, 256>>>((const char*)a, n_rec, rec_kib, 1, b); }, "pattern: 32 adjacent KiB of every 104 KiB record, D 3", moved);
        const size_t n_dense = bytes / (32 * 1024);
        time([&] { readpat<8, 3><<<256, 256>>>((const char*)a, n_dense, 32, 1, b); }, "pattern: records of 32 KiB read whole (dense), one contiguous range per workgroup, D 3", (double)n_dense * 32 * 1024);
        time([&] { readpat<8, 2><<<256, 256>>>((const char*)a, n_dense, 32, 1, b); }, "pattern: the same, D 2", (double)n_dense * 32 * 1024);
    }
    return 0;
}
