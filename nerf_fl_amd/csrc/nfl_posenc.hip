// nfl_posenc.hip -- standalone positional encoding (reference PosEmbedding / BarfPosEmbedding.forward,
// models/nerf.py:19-32, 61-77): (n,3) -> (n, 6N+3) = [x | w_k sin(2^k x) | w_k cos(2^k x) ...].
// The renderer never calls this (it encodes in registers); it exists so that code which calls the
// embedding module directly gets the same arithmetic (exact two-float x/2pi + minimax sine).  HBM-bound.
#include "nfl_render_impl.h"

__global__ __launch_bounds__(256) void nfl_posenc_kernel(const float* x, int n, int nf, const float* w, float* out) {
    const int C = 6 * nf + 3;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)n * C) return;
    const int b = (int)(idx / C), f = (int)(idx % C);
    if (f < 3) {
        out[idx] = x[(size_t)b * 3 + f];
        return;
    }
    const int g = f - 3, k = g / 6, t = (g % 6) / 3, c = g % 3;
    float th, tl;
    nfl_turns(x[(size_t)b * 3 + c], th, tl);
    const float sc = (float)(1 << k);
    const float r = __builtin_amdgcn_fractf(th * sc) + tl * sc + (t ? 0.25f : 0.f);
    out[idx] = (w ? w[k] : 1.f) * nfl_sin_rev(r);
}

extern "C" int nfl_posenc(const float* d_x, int32_t n, int32_t n_freqs, const float* d_w, float* d_out, void* stream) {
    if (!d_x || !d_out || n < 0 || n_freqs < 1 || n_freqs > 16) return NFL_EINVAL;
    if (n == 0) return NFL_OK;
    const long long total = (long long)n * (6 * n_freqs + 3);
    hipLaunchKernelGGL(nfl_posenc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), d_x, n, n_freqs, d_w, d_out);
    return hipGetLastError() == hipSuccess ? NFL_OK : NFL_ELAUNCH;
}

// ---------------------------------------------------------------------------------------------------------
// Ray generation for a frame (reference datasets/ray_utils.py:5-55: get_ray_directions + get_rays): pixel (i, j)
// -> camera direction [(i - cx) / fx, -(j - cy) / fy, -1] (no half-pixel), rotated by c2w[:, :3], normalised;
// origin = c2w[:, 3].  One thread per ray writes the (8)-float row render_rays takes: [o, d, near, far], so an
// eval loop needs only (pose, intrinsics) per frame, not a host-built ray tensor.
__global__ __launch_bounds__(256) void nfl_gen_rays_kernel(const nfl_camera cam, int count, float* rays) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= count) return;
    f4v r0, r1;
    nfl_cam_ray(cam, cam.pix0 + idx, r0, r1);          // shared with the render kernel's camera prologue
    f4v* o = reinterpret_cast<f4v*>(rays + (size_t)idx * 8);
    o[0] = r0;
    o[1] = r1;
}

extern "C" int nfl_gen_rays(const float* h_c2w, float fx, float fy, float cx, float cy, int32_t width, int64_t start,
                            int32_t count, float near, float far, float* d_rays, void* stream) {
    if (!h_c2w || !d_rays || width < 1 || start < 0 || count < 0 || fx == 0.f || fy == 0.f) return NFL_EINVAL;
    if (count == 0) return NFL_OK;
    nfl_camera cam;
    for (int k = 0; k < 12; ++k) cam.c2w[k] = h_c2w[k];
    cam.fx = fx; cam.fy = fy; cam.cx = cx; cam.cy = cy;
    cam.width = width; cam.reserved = 0; cam.pix0 = start; cam.near = near; cam.far = far;
    hipLaunchKernelGGL(nfl_gen_rays_kernel, dim3((count + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), cam,
                       count, d_rays);
    return hipGetLastError() == hipSuccess ? NFL_OK : NFL_ELAUNCH;
}
