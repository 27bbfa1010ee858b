// mfma_shape_probe2.hip -- standalone probe (not part of the library): the two fp16 MFMA shapes under the ISSUE LOAD of
// nfl_render_kernel, not bare.  Per 32-row tile (K = 256, 3 products: 48 x 32x32x16 or 96 x 16x16x32 MFMAs, 1536 MFMA
// cycles either way) the render kernel also issues, in the MFMA shadows: 32 ds_read_b128 (A fragments), 8 LDS-DMA
// pieces of the next chunk, ~90 VALU of the previous tile's activation epilogue (relu, cvt_pk, fma_mix, pack, range
// max) and a barrier + counted vmcnt per tile.  An MFMA holds the SIMD's vector issue for 8 of its cycles whatever its
// shape (MI355X_MICROARCH.md), so the 16x16x32 form leaves 8 x 96 = 768 free issue cycles per tile against 24 x 48 =
// 1152: this probe measures whether the +17 % clock the bare 16x16x32 loop holds (mfma_shape_probe.hip) survives that.
//   hipcc -O3 --offload-arch=gfx950 -o mfma_shape_probe2 mfma_shape_probe2.hip && ./mfma_shape_probe2
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

#define KSTEPS 16
#define TILES 8
#define SLOT (KSTEPS * 2048)
#define LDS_BYTES (3 * SLOT)

// VPK: VALU filler instructions per k-step (the epilogue of the previous tile, spread over this one)
template <int SHAPE, int VPK, int DMA>
__global__ __launch_bounds__(256, 1) void probe(const h8* wsrc, const h8* bsrc, float* out, unsigned long long* clk, int layers) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < LDS_BYTES / 16; i += 256) reinterpret_cast<h8*>(smem)[i] = wsrc[(blockIdx.x * 131 + i) & 65535];
    __syncthreads();
    h8 bh[KSTEPS], bl[KSTEPS];
#pragma unroll
    for (int k = 0; k < KSTEPS; ++k) {
        bh[k] = bsrc[(threadIdx.x * 17 + k * 64 + blockIdx.x) & 65535];
        bl[k] = bsrc[(threadIdx.x * 29 + k * 64 + blockIdx.x + 7) & 65535] * (_Float16)0.001;
    }
    float f0 = lane * 0.001f, f1 = 1.0f, f2 = 0.5f, f3 = 0.25f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float sink = 0.f;
    int slot = 0;
    const char* gsrc = reinterpret_cast<const char*>(wsrc);
    auto valu = [&](int n) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < n; ++i) {
            if ((i & 3) == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f0) : "v"(f1), "v"(f2));
            else if ((i & 3) == 1) asm volatile("v_max_i32 %0, %0, %1" : "+v"(f2) : "v"(f3));
            else if ((i & 3) == 2) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(f3) : "v"(f0), "v"(f2));
            else asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(f1) : "v"(f3));
        }
    };
    auto dma = [&](int k, int nslot) __attribute__((always_inline)) {
        if (DMA && (k & 1) == 0) {
            const unsigned byte = (unsigned)(wave + 4 * (k >> 1)) * 1024u;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc + ((blockIdx.x * 4096u + byte + lane * 16u) & 1048575u)),
                                             (__attribute__((address_space(3))) void*)(smem + nslot * SLOT + byte), 16, 0, 0);
        }
    };
    for (int l = 0; l < layers; ++l) {
#pragma unroll 1
        for (int t = 0; t < TILES; ++t) {
            if (DMA) {
                asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            const char* wl = smem + slot * SLOT + lane * 16;
            const int nslot = slot == 2 ? 0 : slot + 1;
            if constexpr (SHAPE == 0) {
                f16v acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                for (int k = 0; k < KSTEPS; ++k) {
                    const h8 wh = *reinterpret_cast<const h8*>(wl + k * 2048);
                    const h8 wlo = *reinterpret_cast<const h8*>(wl + k * 2048 + 1024);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo, bh[k], acc, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, bl[k], acc, 0, 0, 0);
                    dma(k, nslot);
                    __builtin_amdgcn_sched_barrier(0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, bh[k], acc, 0, 0, 0);
                    valu(VPK);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) sink += acc[r];
            } else {
                f4v a0, a1, a2, a3;
#pragma unroll
                for (int r = 0; r < 4; ++r) a0[r] = a1[r] = a2[r] = a3[r] = 0.f;
                // a 32-row plan tile = two 16-row MFMA tiles (fragment pairs 2k, 2k+1 of a 32-k step), two 16-sample
                // column blocks: per 32-k step 4 fragment reads and 12 MFMAs of 16 cycles
#pragma unroll
                for (int k = 0; k < KSTEPS / 2; ++k) {
                    const h8 wh0 = *reinterpret_cast<const h8*>(wl + (2 * k) * 2048);
                    const h8 wl0 = *reinterpret_cast<const h8*>(wl + (2 * k) * 2048 + 1024);
                    const h8 wh1 = *reinterpret_cast<const h8*>(wl + (2 * k + 1) * 2048);
                    const h8 wl1 = *reinterpret_cast<const h8*>(wl + (2 * k + 1) * 2048 + 1024);
                    a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl0, bh[2 * k], a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl0, bh[2 * k + 1], a1, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh0, bl[2 * k], a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh0, bl[2 * k + 1], a1, 0, 0, 0);
                    dma(2 * k, nslot);
                    __builtin_amdgcn_sched_barrier(0);
                    a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh0, bh[2 * k], a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh0, bh[2 * k + 1], a1, 0, 0, 0);
                    valu(VPK);
                    __builtin_amdgcn_sched_barrier(0);
                    a2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl1, bh[2 * k], a2, 0, 0, 0);
                    a3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl1, bh[2 * k + 1], a3, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    a2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh1, bl[2 * k], a2, 0, 0, 0);
                    a3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh1, bl[2 * k + 1], a3, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    a2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh1, bh[2 * k], a2, 0, 0, 0);
                    a3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh1, bh[2 * k + 1], a3, 0, 0, 0);
                    valu(VPK);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) sink += a0[r] + a1[r] + a2[r] + a3[r];
            }
            slot = nslot;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + threadIdx.x] = sink + f0 + f1 + f2 + f3;
    if (threadIdx.x == 0) {
        clk[2 * blockIdx.x] = t1 - t0;
        clk[2 * blockIdx.x + 1] = r1 - r0;
    }
}

template <int SHAPE, int VPK, int DMA>
static void run(const char* label, const h8* dw, const h8* db, float* dout, unsigned long long* dclk) {
    const int grid = 256, layers = 300;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<SHAPE, VPK, DMA>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float best = 1e9f;
    double cyc = 0, ghz = 0;
    for (int rep = 0; rep < 6; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((probe<SHAPE, VPK, DMA>), dim3(grid), dim3(256), LDS_BYTES, 0, dw, db, dout, dclk, layers);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> hc(grid * 2);
        (void)hipMemcpy(hc.data(), dclk, grid * 16, hipMemcpyDeviceToHost);
        if (rep >= 2 && ms < best) {
            best = ms;
            cyc = (double)hc[0];
            ghz = (double)hc[0] / ((double)hc[1] * 10.0);
        }
    }
    const double macs = (double)grid * 4 * layers * TILES * KSTEPS * 3.0 * 16384.0;
    const double mfma_cycles = (double)layers * TILES * KSTEPS * 3.0 * 32.0;
    printf("%-22s %s  VALU/k-step %2d  DMA %d: %.3f ms  %4.0f TFLOP/s issued  cycles/tile %.0f (MFMA 1536 = %.1f %%)  clock %.2f GHz\n", label,
           SHAPE == 0 ? "32x32x16" : "16x16x32", VPK, DMA, best, 2.0 * macs / (best * 1e-3) / 1e12, cyc / (layers * TILES),
           100.0 * mfma_cycles / cyc, ghz);
}

int main() {
    std::vector<_Float16> hw(65536 * 8), hb(65536 * 8);
    srand(1);
    for (auto& v : hw) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 0.25f);
    for (auto& v : hb) v = (_Float16)(rand() / (float)RAND_MAX);
    h8 *dw, *db;
    float* dout;
    unsigned long long* dclk;
    (void)hipMalloc(&dw, hw.size() * 2);
    (void)hipMalloc(&db, hb.size() * 2);
    (void)hipMalloc(&dout, 256 * 256 * 4);
    (void)hipMalloc(&dclk, 256 * 16);
    (void)hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(db, hb.data(), hb.size() * 2, hipMemcpyHostToDevice);
    run<0, 0, 0>("bare", dw, db, dout, dclk);
    run<1, 0, 0>("bare", dw, db, dout, dclk);
    run<0, 0, 1>("dma only", dw, db, dout, dclk);
    run<1, 0, 1>("dma only", dw, db, dout, dclk);
    run<0, 6, 1>("render-like (6/k-step)", dw, db, dout, dclk);      // ~96 VALU per tile
    run<1, 6, 1>("render-like (6/k-step)", dw, db, dout, dclk);
    run<0, 10, 1>("heavy (10/k-step)", dw, db, dout, dclk);
    run<1, 10, 1>("heavy (10/k-step)", dw, db, dout, dclk);
    return 0;
}
