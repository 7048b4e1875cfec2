// rpt_relaxed.hip — the SAME kernel source (rpt_kernels.hip.h) built a second time with the arithmetic an OpenCL C
// compiler is allowed to use by default, as an opt-in variant (rpt_set_variant 50 / 51), never the default:
//   * FP_CONTRACT ON: a*b + c may become one fma (OpenCL C 1.2 §6.12.2; the reference builds its kernel with no options,
//     CLSetup.cpp:133, so its GPU was free to contract);
//   * x / y within 2.5 ulp and sqrt within 3 ulp (§7.4): v_rcp_f32-based division, raw v_sqrt_f32
//     (-fno-hip-fp32-correctly-rounded-divide-sqrt) instead of the 11- and 17-instruction correctly rounded expansions.
// What it is for: it shows what bit-exactness against the oracle costs (divisions and square roots are a third of the exact
// kernel's vector instructions) and how far a conformant-but-different arithmetic moves the picture — the reference's own
// GPU differs from the exact oracle on 218 pixels of Screenshots/shadows4.png.  Frames of this variant are NOT bit-identical
// to the oracle: tests/test_gpu_relaxed.py reports max |dRGB| and the number of pixels beyond 1e-4 for every configuration,
// and bench.py labels any number measured with it.  The default path and every parity claim use rpt_api.hip's exact build.
#define RPT_RELAXED_FP 1
#define rptd rptd_relaxed            /* its own namespace: both builds of the header live in one library */
#include <hip/hip_runtime.h>
#include "../../include/rpt.h"       /* RPT_TILE_ROWS */
#include "rpt_kernels.hip.h"

namespace rptd_relaxed {
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(5, 5))) void rpt_render_kernel_relaxed_w5(const KernelArgs a) { render_pixel_body<20>(a); }
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(6, 6))) void rpt_render_kernel_relaxed_w6(const KernelArgs a) { render_pixel_body<20>(a); }
}  // namespace rptd_relaxed

// args: the exact build's rptd::KernelArgs, byte for byte (same struct definition, other namespace)
extern "C" int rpt_launch_relaxed_kernel(int waves_per_simd, const void *args, size_t args_bytes, unsigned grid_x, unsigned grid_y, void *stream) {
    if (!args || args_bytes != sizeof(rptd_relaxed::KernelArgs)) return 1;
    rptd_relaxed::KernelArgs a;
    __builtin_memcpy(&a, args, sizeof a);
    if (waves_per_simd == 6) hipLaunchKernelGGL(rptd_relaxed::rpt_render_kernel_relaxed_w6, dim3(grid_x, grid_y), dim3(64), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(rptd_relaxed::rpt_render_kernel_relaxed_w5, dim3(grid_x, grid_y), dim3(64), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
