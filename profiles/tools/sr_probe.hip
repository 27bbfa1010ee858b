// What does v_cvt_sr_f16_f32 (gfx950) do with its random operand?   hipcc --offload-arch=gfx950 sr_probe.hip -o /tmp/sr_probe
// For x = 1 + f * 2^-10 (f = the discarded fraction of an fp16 ulp) it counts how often the conversion rounds up when the
// random word is (a) a full 32-bit hash, (b) only its low 13 bits, (c) only its high 13 bits, (d) bits 13..25.
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__global__ void probe(float x, int mode, unsigned* up, unsigned* hiup) {
    unsigned r = hash(blockIdx.x * 256u + threadIdx.x + 1u);
    if (mode == 1) r &= 0x1fffu;
    if (mode == 2) r &= 0xfff80000u;
    if (mode == 3) r &= (0x1fffu << 13);
    unsigned o = 0u;
    asm volatile("v_cvt_sr_f16_f32 %0, %1, %2" : "+v"(o) : "v"(x), "v"(r));
    if ((o & 0xffffu) != 0x3c00u) atomicAdd(up, 1u);
    unsigned o2 = 0u;
    asm volatile("v_cvt_sr_f16_f32 %0, %1, %2 op_sel:[0,0,1]" : "+v"(o2) : "v"(x), "v"(r));
    if ((o2 >> 16) != 0x3c00u) atomicAdd(hiup, 1u);
    if (blockIdx.x == 0 && threadIdx.x == 0 && mode == 0) printf("  sample: lo-half result %08x  hi-half result %08x\n", o, o2);
}
int main() {
    unsigned *d, h[2];
    hipMalloc(&d, 8);
    const int N = 4096 * 256;
    for (int mode = 0; mode < 4; ++mode)
        for (float f : {0.0f, 0.125f, 0.5f, 0.875f}) {
            hipMemset(d, 0, 8);
            probe<<<4096, 256>>>(1.0f + f / 1024.0f, mode, d, d + 1);
            hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
            printf("mode %d  fraction %.3f  rounded up: lo %.4f  hi %.4f\n", mode, f, double(h[0]) / N, double(h[1]) / N);
        }
    // negative value and a subnormal-range value
    for (float x : {-(1.0f + 0.25f / 1024.0f), 3.0e-6f, 1.0e-8f}) {
        hipMemset(d, 0, 8);
        probe<<<4096, 256>>>(x, 0, d, d + 1);
        hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        printf("x %.9g  'not 0x3c00' count lo %.4f hi %.4f\n", x, double(h[0]) / N, double(h[1]) / N);
    }
    return 0;
}
