// Do the sticky IEEE exception bits (TRAPSTS.EXCP) see an fp32 -> fp16 overflow on gfx950?  (zero-cost range check)
#include <hip/hip_runtime.h>
#include <cstdio>
#define TRAPSTS_EXCP (3 | (0 << 6) | (8 << 11))   // hwreg(HW_REG_TRAPSTS, 0, 9)
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void probe(const float* in, unsigned* out) {
    unsigned t0 = __builtin_amdgcn_s_getreg(TRAPSTS_EXCP);
    float x = in[threadIdx.x & 3];                 // 1.5, 100.25, 3e-6, 1e5
    // (a) in-range conversions only (lanes pick 1.5 / 100.25 / 3e-6): inexact + maybe underflow
    float xa = in[threadIdx.x % 3];
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    h2 p; p[0] = (_Float16)xa; p[1] = (_Float16)(xa * 0.5f);
    unsigned pk = __builtin_bit_cast(unsigned, p);
    asm volatile("" :: "v"(pk));
    unsigned t1 = __builtin_amdgcn_s_getreg(TRAPSTS_EXCP);
    // (b) one lane converts 1e5
    h2 q; q[0] = (_Float16)x; q[1] = (_Float16)xa;
    unsigned pq = __builtin_bit_cast(unsigned, q);
    asm volatile("" :: "v"(pq));
    unsigned t2 = __builtin_amdgcn_s_getreg(TRAPSTS_EXCP);
    // (c) expf overflow / sigmoid of -100
    float e = expf(100.0f * x);
    asm volatile("" :: "v"(e));
    unsigned t3 = __builtin_amdgcn_s_getreg(TRAPSTS_EXCP);
    // (d) MFMA with inf operands
    h8 a, b; for (int j = 0; j < 8; ++j) { a[j] = q[0]; b[j] = (_Float16)0.0f; }
    f16v c = {};
    f16v r = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    asm volatile("" :: "v"(r));
    unsigned t4 = __builtin_amdgcn_s_getreg(TRAPSTS_EXCP);
    if (threadIdx.x == 0) { out[0] = t0; out[1] = t1; out[2] = t2; out[3] = t3; out[4] = t4; out[5] = pq; }
}
int main() {
    float h_in[4] = {1.5f, 100.25f, 3e-6f, 1.0e5f};
    float* d_in; unsigned* d_out; unsigned h_out[8] = {};
    (void)hipMalloc(&d_in, 16); (void)hipMalloc(&d_out, 32);
    (void)hipMemcpy(d_in, h_in, 16, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(d_in, d_out);
    (void)hipMemcpy(h_out, d_out, 32, hipMemcpyDeviceToHost);
    printf("TRAPSTS.EXCP at start %03x | after in-range cvt %03x | after cvt(1e5) %03x | after expf(big) %03x | after mfma(inf*0) %03x | pk %08x\n",
           h_out[0], h_out[1], h_out[2], h_out[3], h_out[4], h_out[5]);
    printf("bits: 0 invalid, 1 input denormal, 2 div0, 3 overflow, 4 underflow, 5 inexact, 6 int div0\n");
    return 0;
}
