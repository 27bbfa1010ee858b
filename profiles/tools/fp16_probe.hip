// fp16 range probe: what do the conversions and the MFMA do beyond 65504 on gfx950?  (hipcc --offload-arch=gfx950 fp16_probe.hip -o fp16_probe)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void probe(const float* in, unsigned* out) {
    const int lane = threadIdx.x;
    float x = in[0], y = in[1];
    _Float16 hx = (_Float16)x;
    unsigned short bits = __builtin_bit_cast(unsigned short, hx);
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    h2 p; p[0] = (_Float16)x; p[1] = (_Float16)y;
    unsigned pk = __builtin_bit_cast(unsigned, p);
    float lo;
    asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(pk), "v"(x));
    // MFMA: A = all (hi of x), B = ones / zeros / mixed
    h8 a, b1, b0;
    for (int j = 0; j < 8; ++j) { a[j] = hx; b1[j] = (_Float16)1.0f; b0[j] = (_Float16)0.0f; }
    f16v c = {};
    f16v r1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b1, c, 0, 0, 0);     // inf * 1 summed
    f16v r0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b0, c, 0, 0, 0);     // inf * 0
    h8 an; for (int j = 0; j < 8; ++j) an[j] = (j & 1) ? hx : (_Float16)(-(float)hx);
    f16v r2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(an, b1, c, 0, 0, 0);    // inf - inf
    if (lane == 0) {
        out[0] = bits; out[1] = pk; out[2] = __builtin_bit_cast(unsigned, lo);
        out[3] = __builtin_bit_cast(unsigned, r1[0]); out[4] = __builtin_bit_cast(unsigned, r0[0]); out[5] = __builtin_bit_cast(unsigned, r2[0]);
        int b = __builtin_bit_cast(int, r2[0]); b = b > 0 ? b : 0; out[6] = (unsigned)b;
    }
}
int main() {
    float h_in[2] = {1.0e5f, -3.0e5f};
    float* d_in; unsigned* d_out; unsigned h_out[8] = {};
    hipMalloc(&d_in, 8); hipMalloc(&d_out, 32);
    hipMemcpy(d_in, h_in, 8, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(d_in, d_out);
    hipMemcpy(h_out, d_out, 32, hipMemcpyDeviceToHost);
    printf("f16(1e5) bits %04x | pk(1e5,-3e5) %08x | lo = x - hi: %08x | mfma inf*1: %08x | inf*0: %08x | inf-inf: %08x | relu_i32(inf-inf): %08x\n",
           h_out[0], h_out[1], h_out[2], h_out[3], h_out[4], h_out[5], h_out[6]);
    return 0;
}
