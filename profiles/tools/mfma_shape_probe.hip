// mfma_shape_probe.hip -- standalone probe (not part of the library): fp16 MFMA throughput and shader clock of the
// two instruction shapes on the operand pattern of nfl_render_kernel: A fragments re-read from LDS with
// ds_read_b128 every k-step, B operands in registers, fp32 accumulators, 3 products per k-step (hi/lo split), one
// wave per SIMD, random data.  The f16x3 forward kernel is clock-limited; this measures what
// the 16x16x32 shape would buy at equal MACs.
//   hipcc -O3 --offload-arch=gfx950 -o mfma_shape_probe mfma_shape_probe.hip && ./mfma_shape_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

#define KSTEPS 16          // k-steps per "row tile" (K = 256)
#define TILES 8            // row tiles per layer
#define LDS_BYTES (KSTEPS * 2048)

// SHAPE 0: v_mfma_f32_32x32x16_f16, one 32-row tile = 16 k-steps x (hi 1 KiB | lo 1 KiB) fragments, 3 MFMAs per k-step
// SHAPE 1: v_mfma_f32_16x16x32_f16, the same 1 KiB fragment = 16 rows x 32 k; two column blocks of 16 samples share
//          it: 6 MFMAs per fragment pair, same MACs per byte read from LDS
template <int SHAPE>
__global__ __launch_bounds__(256, 1) void probe(const h8* wsrc, const h8* bsrc, float* out, unsigned long long* clk, int layers) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < LDS_BYTES / 16; i += 256) reinterpret_cast<h8*>(smem)[i] = wsrc[(blockIdx.x * 131 + i) & 65535];
    __syncthreads();
    h8 bh[KSTEPS], bl[KSTEPS];
#pragma unroll
    for (int k = 0; k < KSTEPS; ++k) {
        bh[k] = bsrc[(threadIdx.x * 17 + k * 64 + blockIdx.x) & 65535];
        bl[k] = bsrc[(threadIdx.x * 29 + k * 64 + blockIdx.x + 7) & 65535] * (_Float16)0.001;
    }
    const char* wl = smem + lane * 16;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float sink = 0.f;
    for (int l = 0; l < layers; ++l) {
#pragma unroll 1
        for (int t = 0; t < TILES; ++t) {
            if constexpr (SHAPE == 0) {
                f16v acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                for (int k = 0; k < KSTEPS; ++k) {
                    const h8 wh = *reinterpret_cast<const h8*>(wl + k * 2048);
                    const h8 wlo = *reinterpret_cast<const h8*>(wl + k * 2048 + 1024);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo, bh[k], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, bl[k], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, bh[k], acc, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) sink += acc[r];
            } else {
                f4v a0, a1;
#pragma unroll
                for (int r = 0; r < 4; ++r) a0[r] = a1[r] = 0.f;
                // two 16-row tiles per iteration so that the MACs per outer iteration match SHAPE 0
#pragma unroll
                for (int half = 0; half < 2; ++half)
#pragma unroll
                    for (int k = 0; k < KSTEPS / 2; ++k) {        // K = 256 = 8 k-steps of 32
                        const int kk = half * (KSTEPS / 2) + k;
                        const h8 wh = *reinterpret_cast<const h8*>(wl + kk * 2048);
                        const h8 wlo = *reinterpret_cast<const h8*>(wl + kk * 2048 + 1024);
                        a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wlo, bh[2 * k], a0, 0, 0, 0);
                        a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wlo, bh[2 * k + 1], a1, 0, 0, 0);
                        a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, bl[2 * k], a0, 0, 0, 0);
                        a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, bl[2 * k + 1], a1, 0, 0, 0);
                        a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, bh[2 * k], a0, 0, 0, 0);
                        a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, bh[2 * k + 1], a1, 0, 0, 0);
                    }
#pragma unroll
                for (int r = 0; r < 4; ++r) sink += a0[r] + a1[r];
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + threadIdx.x] = sink;
    if (threadIdx.x == 0) {
        clk[2 * blockIdx.x] = t1 - t0;
        clk[2 * blockIdx.x + 1] = r1 - r0;
    }
}

int main() {
    const int grid = 256, layers = 400;
    std::vector<_Float16> hw(65536 * 8), hb(65536 * 8);
    srand(1);
    for (auto& v : hw) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 0.25f);
    for (auto& v : hb) v = (_Float16)(rand() / (float)RAND_MAX);
    h8 *dw, *db;
    float* dout;
    unsigned long long* dclk;
    hipMalloc(&dw, hw.size() * 2);
    hipMalloc(&db, hb.size() * 2);
    hipMalloc(&dout, grid * 256 * 4);
    hipMalloc(&dclk, grid * 16);
    hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(db, hb.data(), hb.size() * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int shape = 0; shape < 2; ++shape) {
        float best = 1e9f;
        double cyc = 0, ghz = 0;
        for (int rep = 0; rep < 6; ++rep) {
            hipEventRecord(e0);
            if (shape == 0) hipLaunchKernelGGL(probe<0>, dim3(grid), dim3(256), LDS_BYTES, 0, dw, db, dout, dclk, layers);
            else hipLaunchKernelGGL(probe<1>, dim3(grid), dim3(256), LDS_BYTES, 0, dw, db, dout, dclk, layers);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> hc(grid * 2);
            hipMemcpy(hc.data(), dclk, grid * 16, hipMemcpyDeviceToHost);
            if (rep >= 2 && ms < best) {
                best = ms;
                cyc = (double)hc[0];
                ghz = (double)hc[0] / ((double)hc[1] * 10.0) ;      // s_memrealtime ticks at 100 MHz
            }
        }
        // MFMA MACs per wave: layers * TILES * KSTEPS * 3 products * (32*32*16)
        const double macs = (double)grid * 4 * layers * TILES * KSTEPS * 3.0 * 16384.0;
        const double mfma_cycles = (double)layers * TILES * KSTEPS * 3.0 * 32.0;
        printf("%s: %.3f ms  %.0f TFLOP/s issued (fp16 MFMA)  wave cycles %.0f (MFMA-ideal %.0f = %.1f %%)  shader clock %.2f GHz\n",
               shape == 0 ? "v_mfma_f32_32x32x16_f16" : "v_mfma_f32_16x16x32_f16", best, 2.0 * macs / (best * 1e-3) / 1e12, cyc,
               mfma_cycles, 100.0 * mfma_cycles / cyc, ghz);
    }
    return 0;
}
