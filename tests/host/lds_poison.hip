// Test infrastructure (tests/test_gpu_parity.py::test_query_culling_does_not_depend_on_stale_lds): fills the LDS of every
// CU with NaN bit patterns, so that a kernel launched next which reads LDS another of its waves has not yet written sees
// garbage instead of whatever an earlier workgroup happened to leave there. Built on the GPU box by the test (hipcc).
#include <hip/hip_runtime.h>
__global__ __launch_bounds__(256) void poison(unsigned *sink, unsigned pattern, int words) {
    extern __shared__ unsigned lds[];
    for (int i = threadIdx.x; i < words; i += 256) lds[i] = pattern;
    __syncthreads();
    unsigned acc = 0;
    for (int i = threadIdx.x; i < words; i += 256) acc ^= lds[i];       // keep the stores
    if (acc == 0x12345678u) sink[0] = acc;
}
extern "C" int lds_poison(unsigned pattern, void *stream) {
    static unsigned *sink = nullptr;
    if (!sink && hipMalloc(&sink, 64) != hipSuccess) return 1;
    const int bytes = 160 * 1024;                                      // the whole LDS of a CU, one workgroup per CU
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(poison), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return 2;
    for (int rep = 0; rep < 2; ++rep)
        hipLaunchKernelGGL(poison, dim3(2048), dim3(256), bytes, (hipStream_t)stream, sink, pattern, bytes / 4);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}
