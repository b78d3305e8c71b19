// ORACLE — test infrastructure only (never linked or loaded by the product).
// oracle/_ref/libref_denoise.so = the REFERENCE'S OWN bilateral-denoiser kernels, nerf/renderutils/c_src/denoising.cu (bilateral_denoiser_fwd_kernel :14-71,
// bilateral_denoiser_bwd_kernel :73-130) with the headers they include (common_de.h, math_utils.h, denoising.h, accessor.h), compiled by hipcc for gfx950 from
// where the sources lie under /root/reference — nothing of them is copied into this repository (oracle/Makefile: target _ref). The two compile flags the
// recipe passes, -D__CUDACC__ (selects the device halves of the reference's own #if blocks) and -include hip/hip_runtime.h (float3, blockIdx, ... exactly as
// nvcc pre-includes cuda_runtime.h), are compiler options, not stand-in files. What this file adds is the launch: the host half of the reference's launcher,
// c_src/torch_bindings.cpp:201-246, needs ATen and the CUDA toolkit (absent here), so the kernel arguments are marshalled below the way that code does —
// tensors [n, h, w, c] contiguous, packed_accessor32 (sizes, strides in elements), 8 x 8 x 1 thread blocks, one thread per pixel.
// The kernels run on the GPU: tests/test_gpu_bilateral.py holds csrc/eaw.hip's bilateral kernels AND the CPU oracle to them.
#include REF_DENOISE_CU
#include <cstring>

namespace {
struct AccPod { float* data; int32_t sizes[4]; int32_t strides[4]; };      // the members of PackedTensorAccessor32<float, 4> (accessor.h:258-262), in order
static_assert(sizeof(AccPod) == sizeof(PackedTensorAccessor32<float, 4>), "accessor layout");
void set_acc(PackedTensorAccessor32<float, 4>& a, const float* p, int n, int h, int w, int c) {
    AccPod pod; pod.data = const_cast<float*>(p);
    pod.sizes[0] = n; pod.sizes[1] = h; pod.sizes[2] = w; pod.sizes[3] = c;
    pod.strides[0] = h * w * c; pod.strides[1] = w * c; pod.strides[2] = c; pod.strides[3] = 1;
    std::memcpy(&a, &pod, sizeof(pod));
}
}  // namespace

// col f32[n,h,w,3], nrm f32[n,h,w,3], zdz f32[n,h,w,2] -> out f32[n,h,w,4] (weighted colour sum, weight sum)   (torch_bindings.cpp:201-222)
extern "C" int ref_bilateral_fwd(const float* col, const float* nrm, const float* zdz, float* out4, int n, int h, int w, float sigma, void* stream) {
    BilateralDenoiserParams p;
    std::memset(&p, 0, sizeof(p));
    set_acc(p.col, col, n, h, w, 3); set_acc(p.nrm, nrm, n, h, w, 3); set_acc(p.zdz, zdz, n, h, w, 2); set_acc(p.out, out4, n, h, w, 4);
    p.sigma = sigma;
    dim3 block(8, 8, 1), grid((w - 1) / 8 + 1, (h - 1) / 8 + 1, n);
    bilateral_denoiser_fwd_kernel<<<grid, block, 0, (hipStream_t)stream>>>(p);
    return (int)hipGetLastError();
}
// + out_grad f32[n,h,w,4] -> col_grad f32[n,h,w,3]   (torch_bindings.cpp:224-246)
extern "C" int ref_bilateral_bwd(const float* col, const float* nrm, const float* zdz, const float* out_grad4, float* col_grad, int n, int h, int w, float sigma, void* stream) {
    BilateralDenoiserParams p;
    std::memset(&p, 0, sizeof(p));
    set_acc(p.col, col, n, h, w, 3); set_acc(p.nrm, nrm, n, h, w, 3); set_acc(p.zdz, zdz, n, h, w, 2);
    set_acc(p.out_grad, out_grad4, n, h, w, 4); set_acc(p.col_grad, col_grad, n, h, w, 3);
    p.sigma = sigma;
    dim3 block(8, 8, 1), grid((w - 1) / 8 + 1, (h - 1) / 8 + 1, n);
    bilateral_denoiser_bwd_kernel<<<grid, block, 0, (hipStream_t)stream>>>(p);
    return (int)hipGetLastError();
}
